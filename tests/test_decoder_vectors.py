"""The hand-derived decoder vectors (tests/golden/decoder_vectors.py: UTF-16 surrogate accounting, UTF-8 un-read
rule — what no reference test pins) against BOTH decoders: the oracle's (oracle/sxo.c) and the product's
(sx_codec_core.hpp compiled for the host; the device kernels are compiled from the same source)."""
import ctypes as C

import pytest

import refconfig as rc
import sxo_binding as sxo
import test_replay_core as trc
from golden.decoder_vectors import VECTORS

RESULT = {0: "E", 1: "F", 2: "M"}


class OracleDecoder:
    def __init__(self, enc):
        L = sxo.lib()
        L.sxo_decoder_new.restype = C.c_void_p
        L.sxo_decoder_new.argtypes = [C.c_int]
        L.sxo_decoder_step.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int,
                                       C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.sxo_decoder_free.argtypes = [C.c_void_p]
        self.L, self.d = L, L.sxo_decoder_new(enc)

    def step(self, src, last):
        dst = C.create_string_buffer(256)
        rd, wr = C.c_size_t(), C.c_size_t()
        r = self.L.sxo_decoder_step(self.d, src, len(src), dst, 256, int(last), C.byref(rd), C.byref(wr))
        return RESULT[r], rd.value, wr.value, dst.raw[:wr.value]

    def __del__(self):
        self.L.sxo_decoder_free(self.d)


class ProductDecoder:
    def __init__(self, enc):
        L = trc.load_core()
        L.sxd_decoder_new.restype = C.c_void_p
        L.sxd_decoder_new.argtypes = [C.c_int, C.c_void_p]
        L.sxd_decoder_step.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_int,
                                       C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.sxd_decoder_free.argtypes = [C.c_void_p]
        import stringsext_amd as sx
        t = sx.decoder_table(enc)   # the product's own table (the double-byte decoders read it)
        self.L, self.d = L, L.sxd_decoder_new(enc, C.cast(t[0], C.c_void_p) if t else None)

    def step(self, src, last):
        dst = C.create_string_buffer(256)
        rd, wr = C.c_uint32(), C.c_uint32()
        r = self.L.sxd_decoder_step(self.d, src, len(src), dst, 256, int(last), C.byref(rd), C.byref(wr))
        return RESULT[r], rd.value, wr.value, dst.raw[:wr.value]

    def __del__(self):
        self.L.sxd_decoder_free(self.d)


def drive(dec, calls):
    """the reference's decoder loop (finding_collection.rs:134-143,292-325) over every call's input"""
    out, text = [], b""
    for hexbytes, last, _ in calls:
        rest = bytes.fromhex(hexbytes)
        steps = []
        for _ in range(16):
            r, rd, wr, got = dec.step(rest, last)
            steps.append((r, rd, wr))
            text += got
            rest = rest[rd:]
            if r != "M":
                break
        out.append(steps)
    return out, text


@pytest.mark.parametrize("which", ["oracle", "product"])
@pytest.mark.parametrize("name,enc,calls", VECTORS, ids=[f"{e}:{n}"[:60] for n, e, _ in VECTORS])
def test_decoder_follows_the_hand_derived_vectors(name, enc, calls, which):
    dec = (OracleDecoder if which == "oracle" else ProductDecoder)(rc.ENC_IDS[enc])
    got, text = drive(dec, calls)
    assert got == [steps for _, _, steps in calls], (name, enc)
    # what was written is the text a lenient decoder would give for the well-formed parts
    assert text.decode("utf-8")  is not None


def test_gb18030_token_pending_at_a_buffer_start():
    """Where the token grid of a buffer begins when the carried decoder has bytes pending (what the device replay's look-back
    uses when it walks back to the buffer start): a token boundary of the true grammar — after an error of the pending token the
    bytes given back are decoded in front of the buffer and can take its first byte as their trail (found by tools/gpu_fuzz.py,
    seed 20260930: pending A9 37 C3, then 84)."""
    cases = [("a937c3", "84c39cc3", 1),      # error; 37 and C3 are decoded again, C3 takes 84: the grid starts at 1
             ("81", "30813041", 3),          # the four-byte token finishes after three more bytes
             ("8130", "81304142", 2),
             ("81", "4041", 1),              # a two-byte token
             ("813081", "30414243", 1),
             ("", "81304142", 0)]            # nothing pending
    for pending, nxt, want in cases:
        dec = ProductDecoder(rc.ENC_IDS["gb18030"])
        if pending:
            assert dec.step(bytes.fromhex(pending), False)[:3] == ("E", len(pending) // 2, 0)
        dec.L.sxd_entry_skip.restype = C.c_uint32
        dec.L.sxd_entry_skip.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
        b = bytes.fromhex(nxt)
        assert dec.L.sxd_entry_skip(dec.d, b, len(b)) == want, (pending, nxt)


# ---- the same rules seen through the scan: the position of a finding is where its decoder call began -------------------
def _positions_case():
    """(data, expected findings) written down from the rules in tests/golden/decoder_vectors.py — not from a run."""
    text = "ABCDEFGHIJKL"
    t16 = text.encode("utf-16-le")
    H, H2, L = bytes.fromhex("3dd8"), bytes.fromhex("3cd8"), bytes.fromhex("00de")
    d = bytearray(4096)
    want = []
    d[126:128] = H; d[128:128 + 24] = t16                      # pending high (last unit of window 0) + BMP: both consumed,
    want.append((130, "Exact", text))                           # 'A' is the first output of the call that starts at 130
    d[400:402] = H; d[402:426] = t16                            # high + BMP inside one call: only the high surrogate is consumed
    want.append((402, "Exact", text))
    d[1100:1102] = L; d[1102:1126] = t16                        # lone low: consumed, the next call starts behind it
    want.append((1102, "Exact", text))
    d[1278:1280] = H; d[1280:1282] = H2; d[1282:1284] = L; d[1284:1308] = t16   # pending high, then high: consumed, new one pending;
    want.append((1282, "Exact", text))                          # the pair H2 L (rejected: no astral filter) opens the call at 1282
    d[2000:2002] = H; d[2002:2004] = L; d[2004:2028] = t16      # a real pair in front (rejected char, same call as the window start)
    want.append((1920, "Exact", text))                          # -> the call is the window's: position = window start
    return bytes(d), want


def _scan_positions(run):
    data, want = _positions_case()
    ms = rc.missions(encodings=["utf-16le"], chars_min="10", unicode_block_filter="None")
    out = run(ms, data)
    lines = [l for l in out.decode("utf-8").split("\n")[1:] if l]
    got = []
    for l in lines:
        meta, s = l.split("\t", 1)
        got.append((int(meta[1:].rstrip("+ "), 16), {"<": "Before", " ": "Exact", ">": "After"}[meta[0]], s))
    assert got == want


def test_scan_positions_follow_the_surrogate_rules_oracle_and_host_replay():
    from product_harness import run_cli_product
    _scan_positions(lambda ms, data: sxo.run_cli(ms, [data], radix="x"))
    _scan_positions(lambda ms, data: run_cli_product(ms, [data], radix="x"))


@pytest.mark.gpu
@pytest.mark.parametrize("device_replay", [None, True])
def test_scan_positions_follow_the_surrogate_rules_on_the_gpu(device_replay):
    from product_harness import run_cli_product
    _scan_positions(lambda ms, data: run_cli_product(ms, [data], radix="x", device=0, device_replay=device_replay))
