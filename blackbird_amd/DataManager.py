"""sqlite sink of the self-play path, schema-compatible with the reference's data/blackbird.db
(/root/reference/src/DataManager.py:71-116) so that existing databases keep loading.  Host-side
only; nothing here is on the GPU path."""
import os
import sqlite3

_SCHEMA = """
CREATE TABLE ModelDim(ModelKey INTEGER PRIMARY KEY AUTOINCREMENT, GameType TEXT, Name TEXT, Version INTEGER DEFAULT 1);
CREATE TABLE ConfigurationDim(ConfigurationKey INTEGER PRIMARY KEY AUTOINCREMENT, ConfigJSON TEXT);
CREATE TABLE TrainingStatisticsFact(TrainingStatisticsKey INTEGER PRIMARY KEY AUTOINCREMENT,
    ModelKey INTEGER NOT NULL, OpponentKey INTEGER NOT NULL, Result INTEGER,
    Timestamp DATETIME DEFAULT CURRENT_TIMESTAMP,
    FOREIGN KEY (ModelKey) REFERENCES ModelDim(ModelKey), FOREIGN KEY (OpponentKey) REFERENCES ModelDim(ModelKey));
CREATE TABLE JobQueue(JobKey INTEGER PRIMARY KEY AUTOINCREMENT, QueueTime INTEGER, StartTime INTEGER,
    EndTime INTEGER, RPC INTEGER, PID INTEGER);
CREATE TABLE GameStateFact(GameStateKey INTEGER PRIMARY KEY AUTOINCREMENT, ModelKey INTEGER NOT NULL,
    GameType TEXT NOT NULL, State BYTES NOT NULL, FOREIGN KEY (ModelKey) REFERENCES ModelDim(ModelKey));
INSERT INTO ModelDim(Name, Version) VALUES('RANDOM', 0);
INSERT INTO ModelDim(Name, Version) VALUES('MCTS', 0);
"""


class Connection(object):
    def __init__(self, isLocal=True, directory='data'):
        os.makedirs(directory, exist_ok=True)
        self._conn = sqlite3.connect(os.path.join(directory, 'blackbird.db'))
        self._deferred = False
        self.Cursor = self._conn.cursor()
        have = self.Cursor.execute(
            "SELECT name FROM sqlite_master WHERE type='table' AND name='TrainingStatisticsFact';").fetchone()
        if have is None:
            self.Cursor.executescript(_SCHEMA)

    def _model_key(self, name, version):
        row = self.Cursor.execute(
            "SELECT ModelKey FROM ModelDim WHERE Name = ? AND Version = ? ORDER BY ModelKey DESC LIMIT 1;",
            (name, version)).fetchone()
        return None if row is None else row[0]

    def GetLastVersion(self, gameType, name):
        row = self.Cursor.execute(
            "SELECT Version FROM ModelDim WHERE Name = ? AND GameType = ? ORDER BY Version DESC LIMIT 1;",
            (name, gameType)).fetchone()
        if row is None:
            self.PutModel(gameType, name, 1)
            return 1
        return row[0]

    def PutModel(self, gameType, name, version):
        self.Cursor.execute("INSERT INTO ModelDim(GameType, Name, Version) VALUES(?, ?, ?);", (gameType, name, version))
        self._conn.commit()

    def GetGames(self, name, version):
        key = self._model_key(name, version)
        rows = self.Cursor.execute("SELECT State FROM GameStateFact WHERE ModelKey = ?;", (key,)).fetchall()
        return [r[0] for r in rows]

    def PutGames(self, name, version, gameType, games):
        key = self._model_key(name, version)
        self.Cursor.executemany("INSERT INTO GameStateFact(ModelKey, GameType, State) VALUES(?, ?, ?);",
                                [(key, gameType, g) for g in games])
        if not self._deferred:
            self._conn.commit()

    def Deferred(self):
        """Context manager: the PutGames calls inside share one transaction (one commit at exit instead of one per game).
        GenerateTrainingSamples emits thousands of PutGames back to back once the GPU has finished the games."""
        conn = self

        class _Ctx(object):
            def __enter__(self):
                conn._deferred = True

            def __exit__(self, *exc):
                conn._deferred = False
                conn._conn.commit()
                return False
        return _Ctx()

    def PutTrainingStatistic(self, result, name, version, opName, opVersion=0):
        self.Cursor.execute("INSERT INTO TrainingStatisticsFact(ModelKey, OpponentKey, Result) VALUES(?, ?, ?);",
                            (self._model_key(name, version), self._model_key(opName, opVersion), result))
        self._conn.commit()

    def Close(self):
        if self._conn is not None:
            self._conn.close()
            self._conn = None

    def __del__(self):
        self.Close()

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.Close()
