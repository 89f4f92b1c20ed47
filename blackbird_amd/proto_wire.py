"""Wire format of one training example: proto3 message `State` (/root/reference/src/proto/state.proto:3-8)

    float mctsEval = 1; bytes mctsPolicy = 2; bytes boardEncoding = 3; bytes boardDims = 4; bytes policyDims = 5;

Encoded/decoded by hand so that blobs are byte-identical to what the reference's generated
state_pb2.State.SerializeToString() produces (proto3: fields in number order, zero-valued scalars
and empty bytes omitted), without depending on a protobuf runtime that can still load 2018 stubs.
"""
import struct


def _varint(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _read_varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not (b & 0x80):
            return result, pos
        shift += 7


def encode_state(mctsEval, mctsPolicy, boardEncoding, boardDims, policyDims):
    out = bytearray()
    f = struct.pack('<f', float(mctsEval))
    if f != b'\x00\x00\x00\x00':  # proto3 omits a default-valued float (+0.0); -0.0 is kept
        out += b'\x0d' + f
    for tag, data in ((0x12, mctsPolicy), (0x1a, boardEncoding), (0x22, boardDims), (0x2a, policyDims)):
        data = bytes(data)
        if data:
            out += bytes([tag]) + _varint(len(data)) + data
    return bytes(out)


def encode_states_batch(mctsEval, mctsPolicy, boards):
    """encode_state for K examples at once: mctsEval float32[K], mctsPolicy float64[K, ...], boards int8[K, 1, H, W, C]
    (each example's board is the [1, H, W, C] array AsInputArray returns).  Returns a list of K bytes objects, byte for
    byte what encode_state gives example by example (the dims fields are the int8-cast shapes, Blackbird.py:76-79)."""
    import numpy as np
    ev = np.ascontiguousarray(mctsEval, dtype='<f4')
    K = ev.shape[0]
    pol = np.ascontiguousarray(mctsPolicy, dtype='<f8').reshape(K, -1)
    brd = np.ascontiguousarray(boards, dtype=np.int8).reshape(K, -1)
    bdims = np.array(boards.shape[1:], dtype=np.int64).astype(np.int8).tobytes()
    pdims = np.array(mctsPolicy.shape[1:], dtype=np.int64).astype(np.int8).tobytes()
    pb, bb = pol.shape[1] * 8, brd.shape[1]
    head_p = b'\x12' + _varint(pb) if pb else b''
    head_b = b'\x1a' + _varint(bb) if bb else b''
    tail = (b'\x22' + _varint(len(bdims)) + bdims if bdims else b'') + (b'\x2a' + _varint(len(pdims)) + pdims if pdims else b'')
    L = 5 + len(head_p) + pb + len(head_b) + bb + len(tail)
    m = np.empty((K, L), dtype=np.uint8)
    m[:, 0] = 0x0d
    m[:, 1:5] = ev.view(np.uint8).reshape(K, 4)
    o = 5
    m[:, o:o + len(head_p)] = np.frombuffer(head_p, dtype=np.uint8)
    o += len(head_p)
    m[:, o:o + pb] = pol.view(np.uint8).reshape(K, pb)
    o += pb
    m[:, o:o + len(head_b)] = np.frombuffer(head_b, dtype=np.uint8)
    o += len(head_b)
    m[:, o:o + bb] = brd.view(np.uint8)
    o += bb
    m[:, o:] = np.frombuffer(tail, dtype=np.uint8)
    zero = ev.view('<u4') == 0  # proto3 omits a default-valued float (+0.0 only; -0.0 is kept)
    flat = m.tobytes()
    return [flat[i * L + (5 if zero[i] else 0):(i + 1) * L] for i in range(K)]


def decode_state(blob):
    fields = {1: 0.0, 2: b'', 3: b'', 4: b'', 5: b''}
    pos = 0
    blob = bytes(blob)
    while pos < len(blob):
        key, pos = _read_varint(blob, pos)
        num, wt = key >> 3, key & 7
        if wt == 5:
            val = struct.unpack('<f', blob[pos:pos + 4])[0]
            pos += 4
        elif wt == 2:
            ln, pos = _read_varint(blob, pos)
            val = blob[pos:pos + ln]
            pos += ln
        elif wt == 0:
            val, pos = _read_varint(blob, pos)
        elif wt == 1:
            val = blob[pos:pos + 8]
            pos += 8
        else:
            raise ValueError('unsupported wire type %d' % wt)
        if num in fields:
            fields[num] = val
    return dict(mctsEval=fields[1], mctsPolicy=fields[2], boardEncoding=fields[3], boardDims=fields[4],
                policyDims=fields[5])
