"""Test helper: drive the product (stringsext_amd C-ABI) the way the reference's main::run()
drives FindingCollection::from — file by file, chunk by chunk — and frame the output like the
merger thread does (src/main.rs:116,138)."""
import stringsext_amd as sx
import sxo_binding as sxo


def dbcs_hangover(enc, before, chunk):
    """Double-byte encodings (64 Big5, 65 EUC-JP): how many bytes at the start of `chunk` finish a token that began
    in `before` (the bytes of the stream in front of it) — what the product derives from its carried decoder."""
    lead = {64: lambda b: 0x81 <= b <= 0xFE, 65: lambda b: b in (0x8E, 0x8F) or 0xA1 <= b <= 0xFE,
            66: lambda b: 0x81 <= b <= 0x9F or 0xE0 <= b <= 0xFC, 67: lambda b: 0x81 <= b <= 0xFE}[enc]
    data = before + chunk[:3]
    r = len(before)
    while r > 0 and lead(data[r - 1]):
        r -= 1
    while r < len(before):
        n = 1
        if lead(data[r]):
            n = 3 if (enc == 65 and data[r] == 0x8F and r + 1 < len(data) and 0xA1 <= data[r + 1] <= 0xFE) else 2
        if r + n > len(before):
            over = r + n - len(before)
            # a malformed token gives its last byte back if that byte is ASCII: it is then not part of the hang-over
            last = data[r + n - 1] if r + n - 1 < len(data) else 0x80
            if last < 0x80 and not _dbcs_valid(enc, data[r:r + n]):
                over -= 1
            return over
        r += n
    return 0


def _dbcs_valid(enc, tok):
    codec = {64: "big5hkscs", 65: "euc_jp", 66: "cp932", 67: "cp949"}[enc]
    try:
        bytes(tok).decode(codec)
        return True
    except UnicodeDecodeError:
        return False


def oracle_runs_for_chunk(mdicts, chunk, stream_bytes, before=b""):
    """What stage A must report for this chunk, computed by the oracle's sequential decoder."""
    out = []
    for m in mdicts:
        long_run = max(1, min(m["chars_min_nb"], m["output_line_char_nb_max"]))
        if m["encoding"] in (68, 69):
            # gb18030 / GBK: decode from the last byte in `before` after which the decoder is certainly neutral (neither lead
            # range nor digit), and keep what reaches into the chunk (clipped: stage A's runs are hints for stage B)
            r = len(before)
            while r > 0 and (0x81 <= before[r - 1] <= 0xFE or 0x30 <= before[r - 1] <= 0x39):
                r -= 1
            shift = len(before) - r
            out.append([(max(0, a - shift), b - shift, c) for a, b, c in sxo.runs(m, before[r:] + chunk, min_chars=long_run) if b > shift])
        elif m["encoding"] == 71:
            out.append([])   # ISO-2022-JP: no stage A (one sequential pass on the host, whatever the runs say)
        elif m["encoding"] in (64, 65, 66, 67):
            skip = dbcs_hangover(m["encoding"], before, chunk)
            out.append([(a + skip, b + skip, c) for a, b, c in sxo.runs(m, chunk[skip:], min_chars=long_run)])
        else:
            out.append(sxo.runs(m, chunk, stream_parity=stream_bytes & 1, min_chars=long_run))
    return out


def cut_into_pieces(mdicts, chunk, runs_per_mission):
    """the runs as the device hands them to stage B: cut at the window starts they cross (sx_replay_core.hpp kPieceCont)"""
    import test_replay_core as trc
    core = trc.load_core()
    return [trc.split_runs(core, m, chunk, runs) if m["output_line_char_nb_max"] <= 64 else runs
            for m, runs in zip(mdicts, runs_per_mission)]


def run_cli_product(mdicts, files, radix=None, no_metadata=False, chunk_bytes=None, device=None,
                    generic_kernels=False, subchunk_bytes=0, flush_at_eof=False, record_capacity=0, device_replay=None,
                    pieces=False, replay_threads=0):
    """Whole CLI pass.  device=None: host-only context, runs supplied by the oracle (tests the
    replay stage on CPU).  device=int: the real thing (HIP kernels + replay)."""
    host_only = device is None
    sc = sx.Scanner(mdicts, device=sx.SX_HOST_ONLY if host_only else device, generic_kernels=generic_kernels,
                    subchunk_bytes=subchunk_bytes, record_capacity=record_capacity, device_replay=device_replay,
                    replay_threads=replay_threads)
    out = bytearray(sx.OUTPUT_BOM)
    stream = 0
    before = b""   # the stream in front of the current chunk (decoders persist across files)
    try:
        for fi, data in enumerate(files):
            data = bytes(data)
            step = chunk_bytes or max(len(data), 1)
            assert step % 4096 == 0 or step >= len(data)
            off = 0
            while off < len(data):
                chunk = data[off:off + step]
                last = flush_at_eof and fi == len(files) - 1 and off + len(chunk) == len(data)
                if host_only:
                    runs = oracle_runs_for_chunk(mdicts, chunk, stream, before[-4096:])
                    if pieces:
                        runs = cut_into_pieces(mdicts, chunk, runs)
                    res = sc.replay_runs(chunk, runs, file_id=fi + 1, is_last=last)
                else:
                    res = sc.scan(chunk, file_id=fi + 1, is_last=last)
                out += res.printed(n_inputs=len(files), radix=radix, no_metadata=no_metadata)
                res.free()
                off += len(chunk)
                stream += len(chunk)
                before = (before + chunk)[-8192:]
    finally:
        sc.close()
    out += b"\n"
    return bytes(out)


class ProductScanner:
    """The reference's test-visible surface (ScannerState + FindingCollection::from) on top of
    the product's replay stage, for the unit-test known answers."""

    def __init__(self, mdict, device=None):
        self.m = mdict
        self.device = device
        self.sc = sx.Scanner([mdict], device=sx.SX_HOST_ONLY if device is None else device)
        self.stream = 0
        self.first_byte_position = None
        self.arena = b""

    def scan(self, data, file_id=0, is_last=False):
        data = bytes(data)
        if self.device is None:
            res = self.sc.replay_runs(data, oracle_runs_for_chunk([self.m], data, self.stream), file_id=file_id,
                                      is_last=is_last)
        else:
            res = self.sc.scan(data, file_id=file_id, is_last=is_last)
        self.stream += len(data)
        f = res.findings()
        res.free()
        return f
