"""stringsext_amd — MI355X-native replacement for stringsext's per-Mission byte-stream scan.

This package is only a ctypes shim over the C-ABI in include/stringsext_amd.h
(libstringsext_amd.so: hand-written HIP kernels for gfx950 + a C++ host that
replays the reference's exact window semantics around the runs the device
reports).  Names follow the reference: Mission (src/mission.rs:382-421),
Finding / Precision (src/finding.rs:34-74).  There is no CPU fallback: creating a
Scanner without a HIP device raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SX_LIB") or os.path.join(_HERE, "libstringsext_amd.so")  # SX_LIB: debugging builds

SX_OK, SX_E_INVALID, SX_E_NO_DEVICE, SX_E_HIP, SX_E_NOMEM, SX_E_STATE, SX_E_HALO = 0, -1, -2, -3, -4, -5, -6
SX_HOST_ONLY = -1
SX_OPT_GENERIC_KERNELS, SX_OPT_DEVICE_REPLAY, SX_OPT_HOST_REPLAY = 1, 2, 4
SX_OPT_NO_FUSED_SCAN = 64      # round 6: one scan launch per Mission instead of the fused one (one read of the buffer for several Missions)
SX_OPT_RESULT_ON_DEVICE = 32   # round 5: a string-dense buffer's result stays in HBM (Result.device_segments)
ENC = {"x-user-defined": 0, "utf-8": 1, "utf-16le": 2, "utf-16be": 3, "koi8-r": 16, "ibm866": 17,
       "iso-8859-2": 18, "iso-8859-5": 19, "iso-8859-15": 20, "windows-1251": 21, "windows-1252": 22,
       "iso-8859-3": 23, "iso-8859-4": 24, "iso-8859-6": 25, "iso-8859-7": 26, "iso-8859-8": 27,
       "iso-8859-8-i": 28, "iso-8859-10": 29, "iso-8859-13": 30, "iso-8859-14": 31, "iso-8859-16": 32,
       "koi8-u": 33, "macintosh": 34, "windows-874": 35, "windows-1250": 36, "windows-1253": 37,
       "windows-1254": 38, "windows-1255": 39, "windows-1256": 40, "windows-1257": 41, "windows-1258": 42,
       "x-mac-cyrillic": 43, "big5": 64, "euc-jp": 65, "shift_jis": 66, "euc-kr": 67, "gb18030": 68, "gbk": 69, "replacement": 70}
PRECISION = {0: "Before", 1: "Exact", 2: "After"}

# every symbol include/stringsext_amd.h declares
ABI_VERSION = 4   # SX_ABI_VERSION of include/stringsext_amd.h
EXPORTS = ["sx_abi_version", "sx_create", "sx_destroy", "sx_last_error", "sx_scan", "sx_scan_device", "sx_reset",
           "sx_device_runs", "sx_device_runs_multi", "sx_replay_runs", "sx_scan_shard_device", "sx_scan_shard", "sx_replay_shard_runs",
           "sx_scan_stream", "sx_scan_file", "sx_missions_from_flags", "sx_parse_enc_opt", "sx_encoding_for_label", "sx_encoding_name",
           "sx_decoder_table", "sx_wave_classes", "sx_scan_classifier", "sx_result_segment_packed", "sx_wave_swar", "sx_wave_pair_codes2", "sx_wave_pair_codes", "sx_shard_bounds", "sx_scan_sharded", "sx_shard_splice", "sx_shard_splice_segs", "sx_transport_rccl_id", "sx_transport_rccl_create", "sx_transport_destroy", "sx_transport_last_error", "sx_transport_allgather", "sx_transport_gather",
           "sx_result_count", "sx_result_segments", "sx_result_segment", "sx_result_segment_device", "sx_result_findings", "sx_result_arena",
           "sx_result_free", "sx_print_findings", "sx_get_stats", "sx_free", "sx_fill_background_device",
           "sx_device_alloc", "sx_device_free", "sx_device_upload", "sx_device_download",
           "sx_device_read_bandwidth"]


class Mission(C.Structure):
    """sx_mission — the fields of `Mission` the scan reads."""
    _fields_ = [("mission_id", C.c_uint8), ("encoding", C.c_uint8), ("chars_min_nb", C.c_uint8),
                ("require_same_unicode_block", C.c_uint8), ("grep_char", C.c_int16),
                ("print_encoding_as_ascii", C.c_uint8), ("reserved", C.c_uint8),
                ("output_line_char_nb_max", C.c_uint32), ("af_lo", C.c_uint64), ("af_hi", C.c_uint64),
                ("ubf", C.c_uint64), ("counter_offset", C.c_uint64)]

    @classmethod
    def from_dict(cls, d):
        m = cls()
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

    def to_dict(self):
        return dict(mission_id=self.mission_id, encoding=self.encoding, chars_min_nb=self.chars_min_nb,
                    require_same_unicode_block=bool(self.require_same_unicode_block),
                    grep_char=None if self.grep_char < 0 else self.grep_char,
                    af=(self.af_hi << 64) | self.af_lo, ubf=self.ubf,
                    output_line_char_nb_max=self.output_line_char_nb_max, counter_offset=self.counter_offset,
                    print_encoding_as_ascii=bool(self.print_encoding_as_ascii))


class CliFlags(C.Structure):
    """sx_cli_flags — the option strings of the reference's command line (src/options.rs:46-90)."""
    _fields_ = [("counter_offset", C.c_char_p), ("encodings", C.POINTER(C.c_char_p)), ("n_encodings", C.c_int),
                ("chars_min", C.c_char_p), ("same_unicode_block", C.c_int), ("ascii_filter", C.c_char_p),
                ("unicode_block_filter", C.c_char_p), ("grep_char", C.c_char_p), ("output_line_len", C.c_char_p)]


class EncOpt(C.Structure):
    _fields_ = [("has_name", C.c_int), ("name", C.c_char * 64), ("has_chars_min", C.c_int), ("chars_min", C.c_uint8),
                ("has_af", C.c_int), ("af_lo", C.c_uint64), ("af_hi", C.c_uint64), ("has_ubf", C.c_int), ("ubf", C.c_uint64),
                ("has_grep_char", C.c_int), ("grep_char", C.c_uint8)]


def missions_from_flags(encodings=(), chars_min=None, same_unicode_block=False, ascii_filter=None,
                        unicode_block_filter=None, grep_char=None, output_line_len=None, counter_offset=None):
    """Missions::new (src/mission.rs:514-703) through the C-ABI: option strings -> mission dicts.
    Raises SxError with the reference's message on a bad option."""
    enc = [e.encode() for e in encodings]
    arr = (C.c_char_p * max(1, len(enc)))(*enc)
    b = lambda v: None if v is None else str(v).encode()
    f = CliFlags(b(counter_offset), arr, len(enc), b(chars_min), int(bool(same_unicode_block)), b(ascii_filter),
                 b(unicode_block_filter), b(grep_char), b(output_line_len))
    out = (Mission * 32)()
    n = C.c_int()
    err = C.create_string_buffer(512)
    L = lib()
    L.sx_missions_from_flags.argtypes = [C.POINTER(CliFlags), C.POINTER(Mission), C.c_int, C.POINTER(C.c_int), C.c_char_p, C.c_size_t]
    rc = L.sx_missions_from_flags(C.byref(f), out, 32, C.byref(n), err, 512)
    if rc != SX_OK:
        raise SxError(rc, err.value.decode())
    return [out[i].to_dict() for i in range(n.value)]


def encoding_name(enc):
    """Encoding::name() of an SX_ENC_* id (None if unknown)."""
    L = lib()
    L.sx_encoding_name.argtypes, L.sx_encoding_name.restype = [C.c_uint32], C.c_char_p
    s = L.sx_encoding_name(enc)
    return s.decode() if s else None


def decoder_table(enc):
    """The decoder table of a legacy encoding as a ctypes uint16 array view (None if it has none)."""
    L = lib()
    L.sx_decoder_table.argtypes, L.sx_decoder_table.restype = [C.c_uint32, C.POINTER(C.c_uint64)], C.POINTER(C.c_uint16)
    n = C.c_uint64()
    t = L.sx_decoder_table(enc, C.byref(n))
    return (t, n.value) if n.value else None


def encoding_for_label(label):
    """Encoding::for_label: SX_ENC_* id; -1 not a label; -2 a label of an encoding that is not built in."""
    L = lib()
    L.sx_encoding_for_label.argtypes, L.sx_encoding_for_label.restype = [C.c_char_p], C.c_int
    return L.sx_encoding_for_label(label.encode())


def parse_enc_opt(text):
    """Missions::parse_enc_opt (src/mission.rs:713-749): (name, chars_min, af, ubf, grep_char), None where absent."""
    o = EncOpt()
    err = C.create_string_buffer(512)
    L = lib()
    L.sx_parse_enc_opt.argtypes = [C.c_char_p, C.POINTER(EncOpt), C.c_char_p, C.c_size_t]
    rc = L.sx_parse_enc_opt(text.encode(), C.byref(o), err, 512)
    if rc != SX_OK:
        raise SxError(rc, err.value.decode())
    return (o.name.decode() if o.has_name else None, o.chars_min if o.has_chars_min else None,
            ((o.af_hi << 64) | o.af_lo) if o.has_af else None, o.ubf if o.has_ubf else None,
            o.grep_char if o.has_grep_char else None)


class Finding(C.Structure):
    _fields_ = [("position", C.c_uint64), ("str_off", C.c_uint32), ("str_len", C.c_uint32),
                ("precision", C.c_uint8), ("completes_previous", C.c_uint8), ("mission_id", C.c_uint8),
                ("reserved", C.c_uint8), ("input_file_id", C.c_int16), ("reserved2", C.c_uint16),
                ("slice_index", C.c_uint32)]


class Finding16(C.Structure):   # sx_finding16: a finding as string-dense results cross PCIe
    _fields_ = [("position", C.c_uint64), ("str_off", C.c_uint32), ("str_len", C.c_uint16), ("flags", C.c_uint8), ("mission_id", C.c_uint8)]


class SegmentInfo(C.Structure):   # sx_segment_info
    _fields_ = [("packed", C.c_int32), ("input_file_id", C.c_int32), ("slice_base", C.c_uint32), ("reserved", C.c_uint32),
                ("position0", C.c_uint64 * 256)]


class Run(C.Structure):
    _fields_ = [("start", C.c_uint64), ("end", C.c_uint64), ("chars", C.c_uint64)]


class Stats(C.Structure):
    _fields_ = [("bytes_scanned", C.c_uint64), ("run_records", C.c_uint64), ("replay_bytes", C.c_uint64),
                ("findings", C.c_uint64), ("kernel_ms", C.c_double * 16), ("device_ms", C.c_double),
                ("h2d_ms", C.c_double), ("d2h_ms", C.c_double), ("replay_ms", C.c_double),
                ("total_ms", C.c_double), ("heavy_tiles", C.c_uint64), ("wave_windows", C.c_uint64),
                ("wave_count_ms", C.c_double), ("wave_write_ms", C.c_double), ("rescans", C.c_uint64), ("rescan_ms", C.c_double),
                ("wave_desc_overflows", C.c_uint64), ("seq_pieces", C.c_uint64),
                ("fast_regions", C.c_uint64), ("general_regions", C.c_uint64), ("wave_repairs", C.c_uint64),
                ("fused_ms", C.c_double), ("fused_launches", C.c_uint64), ("fused_mask", C.c_uint64)]


class Options(C.Structure):
    _fields_ = [("subchunk_bytes", C.c_uint32), ("record_capacity", C.c_uint32), ("replay_threads", C.c_uint32),
                ("flags", C.c_uint32)]


class SxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"stringsext_amd error {code}: {msg}")
        self.code = code


_lib = None


def lib():
    """Load libstringsext_amd.so; fail loudly if the HIP extension was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C stringsext_amd/csrc` "
                          "(or __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, u64, cp = C.c_void_p, C.c_uint64, C.c_char_p
    L.sx_abi_version.restype = C.c_int
    if L.sx_abi_version() != ABI_VERSION:   # the ctypes structs below mirror include/stringsext_amd.h of exactly this version
        raise ImportError(f"{LIB_PATH} has ABI version {L.sx_abi_version()}, this binding is written for {ABI_VERSION}: rebuild the library")
    L.sx_create.argtypes = [C.POINTER(vp), C.POINTER(Mission), C.c_int, C.c_int, C.POINTER(Options)]
    L.sx_destroy.argtypes = [vp]
    L.sx_last_error.restype = cp
    L.sx_last_error.argtypes = [vp]
    L.sx_scan.argtypes = [vp, cp, u64, C.c_int, C.c_int, C.POINTER(vp)]
    L.sx_scan_device.argtypes = [vp, vp, u64, C.c_int, C.c_int, C.POINTER(vp)]
    L.sx_reset.argtypes = [vp]
    L.sx_device_runs.argtypes = [vp, C.c_int, vp, u64, C.c_int, u64, C.POINTER(C.POINTER(Run)), C.POINTER(u64)]
    L.sx_device_runs_multi.argtypes = [vp, C.POINTER(C.c_int), C.c_int, vp, u64, C.c_int, C.POINTER(u64), C.POINTER(C.POINTER(Run)), C.POINTER(u64)]
    L.sx_replay_runs.argtypes = [vp, cp, u64, C.c_int, C.c_int, C.POINTER(C.POINTER(Run)), C.POINTER(u64),
                                 C.POINTER(vp)]
    pu64 = C.POINTER(u64)
    L.sx_scan_shard_device.argtypes = [vp, vp, u64, u64, u64, u64, pu64, u64, C.c_int, C.c_int, C.POINTER(vp), pu64]
    L.sx_scan_shard.argtypes = [vp, cp, u64, u64, u64, u64, pu64, u64, C.c_int, C.c_int, C.POINTER(vp), pu64]
    L.sx_replay_shard_runs.argtypes = [vp, cp, u64, u64, u64, u64, pu64, u64, C.c_int, C.POINTER(C.POINTER(Run)), pu64,
                                       C.POINTER(vp), pu64]
    L.sx_result_count.restype = u64
    L.sx_result_count.argtypes = [vp]
    L.sx_result_findings.restype = C.POINTER(Finding)
    L.sx_result_findings.argtypes = [vp]
    L.sx_result_arena.restype = C.POINTER(C.c_uint8)
    L.sx_result_arena.argtypes = [vp, C.POINTER(u64)]
    L.sx_result_free.argtypes = [vp]
    L.sx_result_segments.restype = u64
    L.sx_result_segments.argtypes = [vp]
    L.sx_result_segment.argtypes = [vp, u64, C.POINTER(C.POINTER(Finding)), C.POINTER(u64),
                                    C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(u64)]
    L.sx_print_findings.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.POINTER(C.c_uint8)),
                                    C.POINTER(u64)]
    L.sx_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.sx_free.argtypes = [vp]
    L.sx_fill_background_device.argtypes = [vp, vp, u64, u64, u64]
    L.sx_device_alloc.argtypes = [vp, u64, C.POINTER(vp)]
    L.sx_device_free.argtypes = [vp, vp]
    L.sx_device_upload.argtypes = [vp, vp, cp, u64]
    L.sx_device_download.argtypes = [vp, vp, vp, u64]
    L.sx_device_read_bandwidth.argtypes = [vp, vp, u64, C.c_int, C.POINTER(C.c_double)]
    _lib = L
    return L


class Result:
    """Findings of one sx_scan call, in the reference merger's order (src/main.rs:118-136)."""

    def __init__(self, scanner, handle):
        self._s, self.h = scanner, handle

    def __len__(self):
        return lib().sx_result_count(self.h)

    def raw(self):
        """(findings array bytes, arena bytes) of the whole result as one pair (joined by a copy if it has
        several segments; raises if their strings exceed 4 GiB — read segments() then)."""
        L = lib()
        n = L.sx_result_count(self.h)
        alen = C.c_uint64()
        ap = L.sx_result_arena(self.h, C.byref(alen))
        fp = L.sx_result_findings(self.h)
        if n and not fp:   # flatten failed (sx_result_findings returns NULL): > 4 GiB of strings
            raise SxError(SX_E_INVALID, lib().sx_last_error(self._s.h).decode() or "result too large for one arena: use segments()")
        fb = C.string_at(fp, n * C.sizeof(Finding)) if n else b""
        return fb, (C.string_at(ap, alen.value) if alen.value and ap else b"")

    def segment_pointers(self):
        """[(Finding pointer, n findings, arena pointer, arena bytes)] per segment — no copies."""
        L = lib()
        out = []
        for i in range(L.sx_result_segments(self.h)):
            fp, n, ap, alen = C.POINTER(Finding)(), C.c_uint64(), C.POINTER(C.c_uint8)(), C.c_uint64()
            self._s._chk(L.sx_result_segment(self.h, i, C.byref(fp), C.byref(n), C.byref(ap), C.byref(alen)))
            out.append((fp, n.value, ap, alen.value))
        return out

    def finding_arrays(self):
        """[(Finding pointer, n)] per segment — no copies (the strings stay where they are)."""
        L = lib()
        out = []
        for i in range(L.sx_result_segments(self.h)):
            fp, n, ap, alen = C.POINTER(Finding)(), C.c_uint64(), C.POINTER(C.c_uint8)(), C.c_uint64()
            self._s._chk(L.sx_result_segment(self.h, i, C.byref(fp), C.byref(n), C.byref(ap), C.byref(alen)))
            out.append((fp, n.value))
        return out

    def packed_segments(self):
        """[(packed?, records pointer (Finding16 or Finding), n, arena bytes, SegmentInfo)] — the segments as they are stored"""
        L = lib()
        L.sx_result_segment_packed.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.POINTER(C.c_uint8)),
                                               C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.POINTER(SegmentInfo)]
        out = []
        for i in range(L.sx_result_segments(self.h)):
            fp, n, ap, alen, pk, info = C.c_void_p(), C.c_uint64(), C.POINTER(C.c_uint8)(), C.c_uint64(), C.c_int(), SegmentInfo()
            self._s._chk(L.sx_result_segment_packed(self.h, i, C.byref(fp), C.byref(n), C.byref(ap), C.byref(alen), C.byref(pk), C.byref(info)))
            recs = C.cast(fp, C.POINTER(Finding16 if pk.value else Finding))
            out.append((bool(pk.value), recs, n.value, C.string_at(ap, alen.value) if alen.value else b"", info))
        return out

    def device_segments(self):
        """[(device pointer to the records or None, n, device pointer to the strings, bytes of strings, packed?, SegmentInfo)] —
        SX_OPT_RESULT_ON_DEVICE: the segments that still lie in HBM (None: that segment is in host memory)"""
        L = lib()
        L.sx_result_segment_device.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_void_p),
                                               C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.POINTER(SegmentInfo)]
        out = []
        for i in range(L.sx_result_segments(self.h)):
            fp, n, ap, alen, pk, info = C.c_void_p(), C.c_uint64(), C.c_void_p(), C.c_uint64(), C.c_int(), SegmentInfo()
            self._s._chk(L.sx_result_segment_device(self.h, i, C.byref(fp), C.byref(n), C.byref(ap), C.byref(alen), C.byref(pk), C.byref(info)))
            out.append((fp.value, n.value, ap.value, alen.value, bool(pk.value), info))
        return out

    def segments(self):
        """[(Finding array, n, arena bytes)]: the result as the library holds it (no copy on the C side)."""
        L = lib()
        out = []
        for i in range(L.sx_result_segments(self.h)):
            fp, n, ap, alen = C.POINTER(Finding)(), C.c_uint64(), C.POINTER(C.c_uint8)(), C.c_uint64()
            self._s._chk(L.sx_result_segment(self.h, i, C.byref(fp), C.byref(n), C.byref(ap), C.byref(alen)))
            out.append((fp, n.value, C.string_at(ap, alen.value) if alen.value else b""))
        return out

    def findings(self):
        out = []
        for v, n, arena in self.segments():
            out += [dict(position=v[i].position, precision=PRECISION[v[i].precision],
                         s=arena[v[i].str_off:v[i].str_off + v[i].str_len].decode("utf-8"),
                         completes=bool(v[i].completes_previous), mission_id=v[i].mission_id,
                         file_id=v[i].input_file_id, slice_index=v[i].slice_index) for i in range(n)]
        return out

    def printed(self, n_inputs=1, radix=None, no_metadata=False):
        """Finding::print of every finding (src/finding.rs:112-155), without BOM / final newline."""
        out = C.POINTER(C.c_uint8)()
        n = C.c_uint64()
        self._s._chk(lib().sx_print_findings(self._s.h, self.h, n_inputs, ord(radix) if radix else 0,
                                             int(no_metadata), C.byref(out), C.byref(n)))
        b = C.string_at(out, n.value)
        lib().sx_free(out)
        return b

    def free(self):
        if self.h:
            lib().sx_result_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Scanner:
    """One sx_ctx: N missions bound to one HIP device (device=SX_HOST_ONLY: replay stage only)."""

    def __init__(self, mission_dicts, device=0, subchunk_bytes=0, record_capacity=0, generic_kernels=False,
                 replay_threads=0, device_replay=None, result_on_device=False, fused_scan=True):
        L = lib()
        self.n = len(mission_dicts)
        self._ms = (Mission * self.n)(*[Mission.from_dict(d) for d in mission_dicts])
        flags = SX_OPT_GENERIC_KERNELS if generic_kernels else 0
        if device_replay is True:
            flags |= SX_OPT_DEVICE_REPLAY   # stage B on the device even for small inputs
        elif device_replay is False:
            flags |= SX_OPT_HOST_REPLAY
        if result_on_device:
            flags |= SX_OPT_RESULT_ON_DEVICE
        if not fused_scan:
            flags |= SX_OPT_NO_FUSED_SCAN
        opt = Options(subchunk_bytes, record_capacity, replay_threads, flags)
        self.h = C.c_void_p()
        rc = L.sx_create(C.byref(self.h), self._ms, self.n, device, C.byref(opt))
        if rc != SX_OK:
            self.h = None
            raise SxError(rc, L.sx_last_error(None).decode())

    def _chk(self, rc):
        if rc != SX_OK:
            raise SxError(rc, lib().sx_last_error(self.h).decode())

    def scan(self, data, file_id=-1, is_last=False):
        """sx_scan: replaces the loop src/main.rs:153-168 for one chunk held in host memory."""
        data = bytes(data)
        r = C.c_void_p()
        self._chk(lib().sx_scan(self.h, data, len(data), file_id, int(is_last), C.byref(r)))
        return Result(self, r)

    def scan_device(self, dptr, length, file_id=-1, is_last=False):
        r = C.c_void_p()
        self._chk(lib().sx_scan_device(self.h, dptr, length, file_id, int(is_last), C.byref(r)))
        return Result(self, r)

    def replay_runs(self, data, runs_per_mission, file_id=-1, is_last=False):
        """sx_replay_runs: stage B only; runs_per_mission[m] = [(start, end, chars), ...] sorted."""
        data = bytes(data)
        arrs = [(Run * max(1, len(rs)))(*[Run(*t) for t in rs]) for rs in runs_per_mission]
        ptrs = (C.POINTER(Run) * self.n)(*[C.cast(a, C.POINTER(Run)) for a in arrs])
        ns = (C.c_uint64 * self.n)(*[len(rs) for rs in runs_per_mission])
        r = C.c_void_p()
        self._chk(lib().sx_replay_runs(self.h, data, len(data), file_id, int(is_last), ptrs, ns, C.byref(r)))
        return Result(self, r)

    def scan_shard(self, buf, buf_off, own_lo, own_hi, start_at=None, file_stream_off=0, file_id=-1, reuse_runs=False,
                   runs_per_mission=None, buf_len=None):
        """One rank of a byte-range-sharded scan (sx_scan_shard / _device / sx_replay_shard_runs).
        buf: bytes (host) or a ctypes.c_void_p device pointer (then buf_len is required).
        Returns (Result, end_pos per mission)."""
        sa = (C.c_uint64 * self.n)(*start_at) if start_at is not None else None
        ends = (C.c_uint64 * self.n)()
        r = C.c_void_p()
        if isinstance(buf, C.c_void_p):
            self._chk(lib().sx_scan_shard_device(self.h, buf, buf_off, buf_len, own_lo, own_hi, sa, file_stream_off, file_id,
                                                 int(reuse_runs), C.byref(r), ends))
        else:
            buf = bytes(buf)
            if runs_per_mission is not None:
                arrs = [(Run * max(1, len(rs)))(*[Run(*t) for t in rs]) for rs in runs_per_mission]
                ptrs = (C.POINTER(Run) * self.n)(*[C.cast(a, C.POINTER(Run)) for a in arrs])
                ns = (C.c_uint64 * self.n)(*[len(rs) for rs in runs_per_mission])
                self._chk(lib().sx_replay_shard_runs(self.h, buf, buf_off, len(buf), own_lo, own_hi, sa, file_stream_off,
                                                     file_id, ptrs, ns, C.byref(r), ends))
            else:
                self._chk(lib().sx_scan_shard(self.h, buf, buf_off, len(buf), own_lo, own_hi, sa, file_stream_off, file_id,
                                              int(reuse_runs), C.byref(r), ends))
        return Result(self, r), list(ends)

    def device_runs(self, mission_index, dptr, length, stream_parity=0, min_chars=1, count_only=False):
        """Stage A alone: the long runs of one mission over a device buffer (count_only: just their number)."""
        runs = C.POINTER(Run)()
        n = C.c_uint64()
        self._chk(lib().sx_device_runs(self.h, mission_index, dptr, length, stream_parity, min_chars,
                                       C.byref(runs), C.byref(n)))
        out = n.value if count_only else [(runs[i].start, runs[i].end, runs[i].chars) for i in range(n.value)]
        lib().sx_free(runs)
        return out

    def device_runs_multi(self, mission_indices, dptr, length, stream_parity=0, min_chars=None):
        """Stage A for several missions in one call (sx_device_runs_multi): the missions the fused kernel holds share one
        launch that reads the buffer once.  Returns a list of run lists."""
        n = len(mission_indices)
        idx = (C.c_int * n)(*mission_indices)
        mc = (C.c_uint64 * n)(*(min_chars if min_chars is not None else [1] * n))
        runs = (C.POINTER(Run) * n)()
        cnt = (C.c_uint64 * n)()
        self._chk(lib().sx_device_runs_multi(self.h, idx, n, dptr, length, stream_parity, mc, runs, cnt))
        out = []
        for i in range(n):
            out.append([(runs[i][j].start, runs[i][j].end, runs[i][j].chars) for j in range(cnt[i])])
            lib().sx_free(runs[i])
        return out

    def scan_file(self, path, chunk_bytes=0, file_id=1):
        """Ingest pipeline (sx_scan_file): the file is read, copied to HBM and scanned chunk by chunk,
        overlapped; returns the chunks' Results in input order."""
        results = []
        SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)

        def sink(_user, handle):
            results.append(Result(self, C.c_void_p(handle)))
            return 0
        cb = SINK(sink)
        L = lib()
        L.sx_scan_file.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_int, SINK, C.c_void_p]
        self._chk(L.sx_scan_file(self.h, os.fsencode(path), chunk_bytes, file_id, cb, None))
        return results

    def scan_stream(self, readinto, chunk_bytes=0, file_id=1):
        """sx_scan_stream with a Python reader: readinto(memoryview) -> bytes written (0 at the end)."""
        results = []
        SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)
        READ = C.CFUNCTYPE(C.c_int64, C.c_void_p, C.POINTER(C.c_uint8), C.c_uint64)

        def sink(_user, handle):
            results.append(Result(self, C.c_void_p(handle)))
            return 0

        def read(_user, dst, max_bytes):
            try:
                view = memoryview((C.c_uint8 * max_bytes).from_address(C.addressof(dst.contents))).cast("B")
                return int(readinto(view) or 0)
            except Exception:
                return -1
        cb, rd = SINK(sink), READ(read)
        L = lib()
        L.sx_scan_stream.argtypes = [C.c_void_p, READ, C.c_void_p, C.c_uint64, C.c_int, SINK, C.c_void_p]
        self._chk(L.sx_scan_stream(self.h, rd, None, chunk_bytes, file_id, cb, None))
        return results

    def reset(self):
        self._chk(lib().sx_reset(self.h))

    def stats(self):
        s = Stats()
        self._chk(lib().sx_get_stats(self.h, C.byref(s)))
        return s

    # device memory helpers
    def alloc(self, nbytes):
        p = C.c_void_p()
        self._chk(lib().sx_device_alloc(self.h, nbytes, C.byref(p)))
        return p

    def free(self, dptr):
        self._chk(lib().sx_device_free(self.h, dptr))

    def upload(self, dptr, data):
        data = bytes(data)
        self._chk(lib().sx_device_upload(self.h, dptr, data, len(data)))

    def download(self, dptr, nbytes):
        b = C.create_string_buffer(nbytes)
        self._chk(lib().sx_device_download(self.h, b, dptr, nbytes))
        return b.raw

    def fill_background(self, dptr, first_index, nbytes, seed=0x5EED5EED5EED5EED):
        self._chk(lib().sx_fill_background_device(self.h, dptr, first_index, nbytes, seed))

    def read_bandwidth(self, dptr, nbytes, repeats=5):
        g = C.c_double()
        self._chk(lib().sx_device_read_bandwidth(self.h, dptr, nbytes, repeats, C.byref(g)))
        return g.value

    def close(self):
        if self.h:
            lib().sx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


OUTPUT_BOM = b"\xEF\xBB\xBF"
