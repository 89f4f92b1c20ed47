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
