"""ctypes binding of the ORACLE (oracle/libsxo.so). Test infrastructure only."""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None


class Mission(C.Structure):
    _fields_ = [("mission_id", C.c_uint8), ("encoding", C.c_uint8), ("chars_min_nb", C.c_uint8),
                ("require_same_unicode_block", C.c_uint8), ("grep_char", C.c_int16),
                ("print_encoding_as_ascii", C.c_uint8), ("_pad", C.c_uint8),
                ("output_line_char_nb_max", C.c_uint32), ("af_lo", C.c_uint64), ("af_hi", C.c_uint64),
                ("ubf", C.c_uint64), ("counter_offset", C.c_uint64)]


class Finding(C.Structure):
    _fields_ = [("position", C.c_uint64), ("s_off", C.c_uint32), ("s_len", C.c_uint32),
                ("precision", C.c_uint8), ("completes_previous", C.c_uint8), ("mission_id", C.c_uint8),
                ("input_file_id", C.c_int16)]


class FC(C.Structure):
    _fields_ = [("v", C.POINTER(Finding)), ("n", C.c_size_t), ("cap", C.c_size_t),
                ("arena", C.POINTER(C.c_uint8)), ("arena_len", C.c_size_t), ("arena_cap", C.c_size_t),
                ("first_byte_position", C.c_uint64), ("str_buf_overflow", C.c_int)]


class SplitResult(C.Structure):
    _fields_ = [("s_off", C.c_uint32), ("s_len", C.c_uint32), ("completes_previous", C.c_uint8),
                ("is_maybe_cut", C.c_uint8), ("to_be_filtered_again", C.c_uint8), ("min_ok", C.c_uint8),
                ("grep_ok", C.c_uint8)]


class File(C.Structure):
    _fields_ = [("data", C.c_char_p), ("len", C.c_size_t)]


class RunRec(C.Structure):
    _fields_ = [("start", C.c_uint64), ("end", C.c_uint64), ("chars", C.c_uint64)]


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(os.path.join(ROOT, "oracle", "libsxo.so"))
        _lib.sxo_state_new.restype = C.c_void_p
        _lib.sxo_state_new.argtypes = [C.POINTER(Mission)]
        _lib.sxo_state_free.argtypes = [C.c_void_p]
        _lib.sxo_state_consumed_bytes.restype = C.c_uint64
        _lib.sxo_state_consumed_bytes.argtypes = [C.c_void_p]
        _lib.sxo_state_maybe_cut.argtypes = [C.c_void_p]
        _lib.sxo_state_leftover.restype = C.POINTER(C.c_uint8)
        _lib.sxo_state_leftover.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        _lib.sxo_fc_init.argtypes = [C.POINTER(FC)]
        _lib.sxo_fc_free.argtypes = [C.POINTER(FC)]
        _lib.sxo_scan_slice.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t, C.c_int, C.POINTER(FC)]
        _lib.sxo_split_str.restype = C.c_size_t
        _lib.sxo_split_str.argtypes = [C.c_char_p, C.c_size_t, C.c_uint8, C.c_int, C.c_int, C.c_int, C.c_uint64,
                                       C.c_uint64, C.c_uint64, C.c_int, C.c_size_t, C.POINTER(SplitResult),
                                       C.c_size_t]
        _lib.sxo_run.argtypes = [C.POINTER(Mission), C.c_int, C.POINTER(File), C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
        _lib.sxo_run_count.argtypes = [C.POINTER(Mission), C.c_int, C.POINTER(File), C.c_int,
                                       C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        _lib.sxo_free.argtypes = [C.c_void_p]
        _lib.sxo_runs.restype = C.c_size_t
        _lib.sxo_runs.argtypes = [C.POINTER(Mission), C.c_char_p, C.c_size_t, C.c_int, C.c_uint64,
                                  C.POINTER(RunRec), C.c_size_t]
        _lib.sxo_fill_background.argtypes = [C.c_void_p, C.c_uint64, C.c_size_t, C.c_uint64]
        _lib.sxo_encoding_name.restype = C.c_char_p
        _lib.sxo_encoding_name.argtypes = [C.c_int]
    return _lib


def to_mission(d):
    m = Mission()
    m.mission_id = d["mission_id"]
    m.encoding = d["encoding"]
    m.chars_min_nb = d["chars_min_nb"]
    m.require_same_unicode_block = 1 if d["require_same_unicode_block"] else 0
    m.grep_char = -1 if d["grep_char"] is None else d["grep_char"]
    m.print_encoding_as_ascii = 1 if d["print_encoding_as_ascii"] else 0
    m.output_line_char_nb_max = d["output_line_char_nb_max"]
    m.af_lo = d["af"] & 0xFFFFFFFFFFFFFFFF
    m.af_hi = d["af"] >> 64
    m.ubf = d["ubf"]
    m.counter_offset = d["counter_offset"]
    return m


PRECISION = {0: "Before", 1: "Exact", 2: "After"}


class Scanner:
    """ScannerState + FindingCollection::from, as the reference's tests use them."""

    def __init__(self, mdict):
        self.m = to_mission(mdict)
        self.h = lib().sxo_state_new(C.byref(self.m))
        self.fc = FC()
        lib().sxo_fc_init(C.byref(self.fc))

    def scan(self, data, file_id=0, is_last=False):
        lib().sxo_scan_slice(self.h, file_id, bytes(data), len(data), 1 if is_last else 0, C.byref(self.fc))
        arena = bytes(self.fc.arena[:self.fc.arena_len]) if self.fc.arena_len else b""
        out = []
        for i in range(self.fc.n):
            f = self.fc.v[i]
            out.append(dict(position=f.position, precision=PRECISION[f.precision],
                            s=arena[f.s_off:f.s_off + f.s_len].decode("utf-8"),
                            completes=bool(f.completes_previous), file_id=f.input_file_id))
        self.arena = arena
        return out

    @property
    def first_byte_position(self):
        return self.fc.first_byte_position

    @property
    def consumed_bytes(self):
        return lib().sxo_state_consumed_bytes(self.h)

    @property
    def maybe_cut(self):
        return bool(lib().sxo_state_maybe_cut(self.h))

    @property
    def leftover(self):
        n = C.c_size_t()
        p = lib().sxo_state_leftover(self.h, C.byref(n))
        return bytes(p[:n.value]).decode("utf-8")

    def __del__(self):
        try:
            lib().sxo_fc_free(C.byref(self.fc))
            lib().sxo_state_free(self.h)
        except Exception:
            pass


def split_str(inp, chars_min_nb, same_block, last_cut, invalid_after, af, ubf, grep, q):
    if isinstance(inp, str):
        inp = inp.encode("utf-8")
    buf = (SplitResult * 256)()
    n = lib().sxo_split_str(inp, len(inp), chars_min_nb, int(same_block), int(last_cut), int(invalid_after),
                            af & 0xFFFFFFFFFFFFFFFF, af >> 64, ubf, -1 if grep is None else grep, q, buf, 256)
    return [dict(s=inp[r.s_off:r.s_off + r.s_len].decode("utf-8"), completes=bool(r.completes_previous),
                 maybe_cut=bool(r.is_maybe_cut), again=bool(r.to_be_filtered_again), min_ok=bool(r.min_ok),
                 grep_ok=bool(r.grep_ok)) for r in buf[:n]]


def run_cli(mdicts, files, radix=None, no_metadata=False, flush_at_eof=False):
    ms = (Mission * len(mdicts))(*[to_mission(d) for d in mdicts])
    keep = [bytes(f) for f in files]
    fs = (File * len(keep))(*[File(k, len(k)) for k in keep])
    out = C.POINTER(C.c_uint8)()
    n = C.c_size_t()
    lib().sxo_run(ms, len(mdicts), fs, len(keep), ord(radix) if radix else 0, int(no_metadata),
                  int(flush_at_eof), C.byref(out), C.byref(n))
    res = bytes(out[:n.value])
    lib().sxo_free(out)
    return res


def run_count(mdicts, files):
    ms = (Mission * len(mdicts))(*[to_mission(d) for d in mdicts])
    keep = [bytes(f) for f in files]
    fs = (File * len(keep))(*[File(k, len(k)) for k in keep])
    a, b = C.c_uint64(), C.c_uint64()
    lib().sxo_run_count(ms, len(mdicts), fs, len(keep), C.byref(a), C.byref(b))
    return a.value, b.value


def runs(mdict, data, stream_parity=0, min_chars=1, cap=1 << 20):
    m = to_mission(mdict)
    buf = (RunRec * cap)()
    n = lib().sxo_runs(C.byref(m), bytes(data), len(data), stream_parity, min_chars, buf, cap)
    assert n <= cap
    return [(r.start, r.end, r.chars) for r in buf[:n]]


def background(first, length, seed=0x5EED5EED5EED5EED):
    b = C.create_string_buffer(length)
    lib().sxo_fill_background(b, first, length, seed)
    return b.raw
