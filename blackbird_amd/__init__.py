"""blackbird_amd -- MI355X-native batched self-play / MCTS engine behind BlackBird's Python API.

Host code is Python; all compute goes through the C ABI of libblackbird_hip.so
(include/blackbird_hip.h) via ctypes.  Importing the package does not load the library; the
first operation that needs it does, and fails loudly if it is missing (there is no CPU fallback).
"""
__version__ = "0.1.0"
