"""Bitstream container of DCVC-RT: SPS / I / P NAL units with variable-length integers, byte-compatible
with the reference's ``src/utils/stream_helper.py`` (write_uint_adaptive :68-89, read_uint_adaptive
:92-105, NalType :108, SPSHelper :114-145, write_sps :148-162, read_header :165-184,
read_sps_remaining :187-195, write_ip :198-209, read_ip_remaining :212-217), so ``.bin`` files are
interchangeable with the reference's encoder / decoder.  Same function names for drop-in use; `f` is any
binary file-like object (io.BytesIO, open(..., 'wb')).

Layout:  NAL header byte = type(4 bits) | sps_id(4 bits)
  SPS :  header, height (varint), width (varint), flags = ec_part << 2 | use_ada_i
  I/P :  header, qp (1 byte), payload length (varint), payload (the rANS stream of the frame)
varint:  0xxxxxxx                      value < 2**7
         10xxxxxx xxxxxxxx             value < 2**14   (big endian)
         11xxxxxx + 3 bytes            value < 2**30   (big endian)
"""
import enum


class NalType(enum.IntEnum):
    NAL_SPS = 0
    NAL_I = 1
    NAL_P = 2


def write_uint_adaptive(f, value):
    if value < 0 or value >= (1 << 30):
        raise ValueError(f"varint out of range: {value}")
    if value < (1 << 7):
        data = bytes((value,))
    elif value < (1 << 14):
        data = bytes((0x80 | (value >> 8), value & 0xff))
    else:
        data = bytes((0xc0 | (value >> 24), (value >> 16) & 0xff, (value >> 8) & 0xff, value & 0xff))
    f.write(data)
    return len(data)


def _byte(f):
    b = f.read(1)
    if len(b) != 1:
        raise EOFError("truncated DCVC-RT stream")
    return b[0]


def read_uint_adaptive(f):
    first = _byte(f)
    if first < 0x80:
        return first
    if (first >> 6) == 0x02:
        return ((first & 0x3f) << 8) | _byte(f)
    rest = f.read(3)
    if len(rest) != 3:
        raise EOFError("truncated DCVC-RT stream")
    return ((first & 0x3f) << 24) | (rest[0] << 16) | (rest[1] << 8) | rest[2]


class SPSHelper:
    """Deduplicates sequence parameter sets (at most 16 ids), reference stream_helper.py:114-145."""
    _KEYS = ("height", "width", "use_ada_i", "ec_part")

    def __init__(self):
        self.spss = []

    def get_sps_id(self, target_sps):
        for sps in self.spss:
            if all(sps[k] == target_sps[k] for k in self._KEYS):
                return sps["sps_id"], False
        new_id = max((s["sps_id"] for s in self.spss), default=-1) + 1
        if new_id > 15:
            raise ValueError("more than 16 distinct SPS in one stream")
        sps = dict(target_sps, sps_id=new_id)
        self.spss.append(sps)
        return new_id, True

    def add_sps_by_id(self, sps):
        for i, s in enumerate(self.spss):
            if s["sps_id"] == sps["sps_id"]:
                self.spss[i] = dict(sps)
                return
        self.spss.append(dict(sps))

    def get_sps_by_id(self, sps_id):
        for s in self.spss:
            if s["sps_id"] == sps_id:
                return s
        return None


def write_sps(f, sps):
    if not (0 <= sps["sps_id"] < 16 and sps["use_ada_i"] in (0, 1) and sps["ec_part"] in (0, 1)):
        raise ValueError(f"bad SPS {sps}")
    f.write(bytes(((int(NalType.NAL_SPS) << 4) | sps["sps_id"],)))
    n = 1 + write_uint_adaptive(f, sps["height"]) + write_uint_adaptive(f, sps["width"])
    f.write(bytes(((sps["ec_part"] << 2) | sps["use_ada_i"],)))
    return n + 1


def read_header(f):
    flag = _byte(f)
    nal_type = flag >> 4
    header = {"nal_type": NalType(nal_type)}
    header["sps_id"] = flag & 0x0f
    return header


def read_sps_remaining(f, sps_id):
    height = read_uint_adaptive(f)
    width = read_uint_adaptive(f)
    flag = _byte(f)
    return {"sps_id": sps_id, "height": height, "width": width, "ec_part": (flag >> 2) & 1, "use_ada_i": flag & 1}


def write_ip(f, is_i_frame, sps_id, qp, bit_stream):
    if not 0 <= qp < 256:
        raise ValueError(f"qp {qp} out of range")
    f.write(bytes(((int(NalType.NAL_I if is_i_frame else NalType.NAL_P) << 4) | sps_id, qp)))
    n = 2 + write_uint_adaptive(f, len(bit_stream))
    f.write(bit_stream)
    return n + len(bit_stream)


def read_ip_remaining(f):
    qp = _byte(f)
    length = read_uint_adaptive(f)
    bit_stream = f.read(length)
    if len(bit_stream) != length:
        raise EOFError("truncated DCVC-RT frame payload")
    return qp, bit_stream


class StreamWriter:
    """What test_video.py:166,216-224 does per frame: SPS dedup + NAL writing; returns bytes written."""

    def __init__(self, f):
        self.f = f
        self.sps_helper = SPSHelper()

    def write_frame(self, height, width, use_two_entropy_coders, pkt):
        sps = {"sps_id": -1, "height": height, "width": width, "ec_part": 1 if use_two_entropy_coders else 0,
               "use_ada_i": pkt.use_ada_i}
        sps_id, is_new = self.sps_helper.get_sps_id(sps)
        sps["sps_id"] = sps_id
        n = write_sps(self.f, sps) if is_new else 0
        return n + write_ip(self.f, pkt.is_i, sps_id, pkt.qp, pkt.bit_stream)


class StreamReader:
    """test_video.py:265-276: yields (sps, is_i_frame, qp, payload) per frame."""

    def __init__(self, f):
        self.f = f
        self.sps_helper = SPSHelper()

    def read_frame(self):
        header = read_header(self.f)
        while header["nal_type"] == NalType.NAL_SPS:
            self.sps_helper.add_sps_by_id(read_sps_remaining(self.f, header["sps_id"]))
            header = read_header(self.f)
        sps = self.sps_helper.get_sps_by_id(header["sps_id"])
        if sps is None:
            raise ValueError(f"frame refers to unknown SPS {header['sps_id']}")
        qp, payload = read_ip_remaining(self.f)
        return sps, header["nal_type"] == NalType.NAL_I, qp, payload
