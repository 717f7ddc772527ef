"""One randomized parity case from a seed (shared by gpu_fuzz.py and gpu_repro.py)."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import refconfig as rc
from test_host_logic import soup, synth
from test_gpu_parity import dense

ENCS = ["utf-8", "ascii", "utf-16le", "utf-16be", "koi8-r", "ibm866", "windows-1252", "iso-8859-5", "x-user-defined", "big5", "euc-jp", "shift_jis", "euc-kr", "gbk", "gb18030", "iso-2022-jp"]
ENCS_MORE = ["iso-8859-7", "windows-1255", "windows-874", "koi8-u", "macintosh", "iso-8859-6", "windows-1257", "x-mac-cyrillic"]
AFS = [None, "All", "All-Ctrl", "All-Ctrl+Wsp", "None", "Wsp", "0x7ffffffe000000007ffffffe00000000"]
UBFS = [None, "African", "All", "Common", "Cyrillic", "Latin", "Asian", "Uncommon", "None", "Hebrew", "Cjk", "Hangul", "Kana", "0x0000fffe00000000", "0x00003ffcfffffffc"]
# alternative paths behind environment switches, read by the library at call time
SWITCH_SETS = [{}, {}, {}, {"SX_RESULT_ON_DEVICE": "1"}, {"SX_RESULT_ON_DEVICE": "1", "SX_WAVE_REPLAY": "1"}, {"SX_RESULT_ON_DEVICE": "1", "SX_DEVICE_JOIN_MIN": "1"}, {"SX_NO_REPLAY_CACHE": "1"}, {"SX_HOST_STITCH": "1"}, {"SX_NO_REPLAY_SKIP": "1"}, {"SX_REPLAY_CACHE_MIB": "0"},
               {"SX_REGION_CAP": "2"}, {"SX_REGION_CAP": "0"}, {"SX_DEVICE_JOIN_MIN": "1"}, {"SX_HOST_MERGE": "1"},
               {"SX_NO_REPLAY_CACHE": "1", "SX_NO_REPLAY_SKIP": "1"}, {"SX_HOST_STITCH": "1", "SX_DEVICE_JOIN_MIN": "1"},
               {"SX_REGION_CAP": "1"}, {"SX_REGION_CAP": "2", "SX_NO_LARGE_REGIONS": "1"}, {"SX_REGION_CAP": "1", "SX_DEVICE_JOIN_MIN": "1"},
               {"SX_STITCH_BLOCK": "512"}, {"SX_STITCH_BLOCK": "3"}, {"SX_MAX_REGION_WINDOWS": "2"}, {"SX_MAX_REGION_WINDOWS": "64"},
               {"SX_SLABS": "3"}, {"SX_SLABS": "8"}, {"SX_SLABS": "5", "SX_MAX_REGION_WINDOWS": "2"}, {"SX_SLABS": "2", "SX_DEVICE_JOIN_MIN": "1"},
               {"SX_DEFER_MIN_BYTES": "1"}, {"SX_DEFER_MIN_BYTES": "1", "SX_MERGE_PART_FINDINGS": "2000"}, {"SX_MERGE_PART_FINDINGS": "3000"},
               # a buffer in pieces scanned one after the other, the copy of a piece's merged findings next to the following piece
               {"SX_SEQ_PIECE_KIB": "8"}, {"SX_SEQ_PIECE_KIB": "16", "SX_DEFER_MIN_BYTES": "1"}, {"SX_SEQ_PIECE_KIB": "12", "SX_DEFER_MIN_BYTES": "1", "SX_WAVE_REPLAY": "1"},
               # the wave-cooperative stage B (sx_wave.cpp): forced on for every covered Mission, with odd wavefront / slab geometries,
               # with and without stage A in front of it, with its way back to the lane-per-region path, and forced off
               {"SX_WAVE_REPLAY": "1"}, {"SX_WAVE_REPLAY": "1"}, {"SX_WAVE_REPLAY": "1"}, {"SX_WAVE_REPLAY": "1", "SX_WAVE_BATCHES": "1"},
               {"SX_WAVE_REPLAY": "1", "SX_WAVE_BATCHES": "2", "SX_WAVE_SLABS": "3"}, {"SX_WAVE_REPLAY": "1", "SX_WAVE_SLABS": "5", "SX_WAVE_KEEP_SCAN": "1"},
               {"SX_WAVE_REPLAY": "1", "SX_DEFER_MIN_BYTES": "1"}, {"SX_WAVE_REPLAY": "1", "SX_WAVE_FAIL": "1"}, {"SX_WAVE_REPLAY": "0"},
               {"SX_WAVE_REPLAY": "1", "SX_WAVE_DESC": "0"}, {"SX_WAVE_REPLAY": "1", "SX_WAVE_DESC_CAP": "7"},
               {"SX_WAVE_REPLAY": "1", "SX_WAVE_THREADS": "0"}, {"SX_WAVE_REPLAY": "1", "SX_WAVE_THREADS": "0", "SX_DEFER_MIN_BYTES": "1"},
               {"SX_WAVE_REPLAY": "1", "SX_WAVE_DESC_CAP": "40", "SX_WAVE_SLABS": "4"}, {"SX_WAVE_REPLAY": "1", "SX_WAVE_DESC": "0", "SX_WAVE_BATCHES": "2"},
               {"SX_WAVE_REPLAY": "1", "SX_WAVE_LUT": "1"}, {"SX_WAVE_REPLAY": "1", "SX_WAVE_LUT": "1", "SX_WAVE_BATCHES": "1"},
               # round 5: the lane-per-region path with and without its fast pre-pass (sx_replay_dev.hip replay_fast_kernel), on string-dense input too
               {"SX_FAST_REPLAY": "0"}, {"SX_WAVE_REPLAY": "0"}, {"SX_WAVE_REPLAY": "0"}, {"SX_WAVE_REPLAY": "0", "SX_SLABS": "3"},
               {"SX_WAVE_REPLAY": "0", "SX_REPLAY_CACHE_MIB": "1"}, {"SX_WAVE_REPLAY": "0", "SX_STITCH_BLOCK": "3"},
               {"SX_WAVE_REPLAY": "0", "SX_FAST_REPLAY": "0", "SX_STITCH_BLOCK": "3"}, {"SX_WAVE_REPLAY": "0", "SX_NO_PIECES": "1"}]
ALL_SWITCHES = sorted({k for s in SWITCH_SETS for k in s} | {"SX_FUSED", "SX_FUSED_PREFILTER", "SX_PIECE_MIB"})


def make(case_seed):
    r = random.Random(case_seed)
    encs = []
    for _ in range(r.randrange(1, 4)):
        e = r.choice(ENCS)
        if r.random() < 0.3:
            e += "," + r.choice(["", "2", "5", "12"]) + "," + (r.choice(AFS) or "") + "," + (r.choice(UBFS) or "")
        encs.append(e)
    kw = dict(encodings=encs, chars_min=r.choice([None, "0", "1", "2", "4", "7", "10", "20", "70"]),
              output_line_len=r.choice([None, None, "6", "8", "10", "30", "64", "100"]),
              ascii_filter=r.choice(AFS), unicode_block_filter=r.choice(UBFS),
              grep_char=r.choice([None, None, None, "47", "0x65", "32"]), same_unicode_block=r.random() < 0.2,
              counter_offset=r.choice([None, None, "1000", "0x10"]))
    ms = rc.missions(**kw)
    kind = r.choice(["synth", "synth_dense", "soup", "dense", "text", "multi", "cjk"])
    size = r.choice([5000, 70_000, 300_000, 1_200_000])
    if kind == "synth": files = [synth(r, size, 1 / 500)]
    elif kind == "synth_dense": files = [synth(r, size, 1 / 60)]
    elif kind == "soup": files = [soup(r, min(size, 200_000))]
    elif kind == "dense": files = [dense(r, size, r.choice([2, 20, 200]), "abcdefgh XYZ019_-éжЖдяבשλ€😀")]
    elif kind == "text": files = [("The quick brown fox — Ünïcödé ßtring, доброе утро, שלום עולם. " * (size // 60 + 1)).encode(r.choice(["utf-8", "utf-16-le", "koi8-r"]), errors="replace")[:size]]
    elif kind == "cjk":
        from test_dbcs import soup as dbcs_soup
        from test_iso2022jp import soup as iso_soup
        which = r.choice(["big5", "euc-jp", "shift_jis", "euc-kr", "gbk", "gb18030", "iso-2022-jp"])
        files = [iso_soup(r, min(size, 300_000)) if which == "iso-2022-jp" else dbcs_soup(which, r, min(size, 300_000))]
    else: files = [synth(r, r.randrange(1, 20000), 1 / 100) for _ in range(r.randrange(2, 6))] + [b""]
    case = dict(kw=kw, missions=ms, kind=kind, size=size, files=files, chunk=r.choice([None, None, 4096, 16384, 65536]),
                flush=r.random() < 0.3, sub=r.choice([0, 0, 1024, 4096]), replay=r.choice([None, None, True, False]),
                generic=r.random() < 0.2)
    case["switches"] = r.choice(SWITCH_SETS)
    if r.random() < 0.15:  # one of the other WHATWG single-byte tables instead of the first mission's encoding
        kw["encodings"][0] = r.choice(ENCS_MORE) + kw["encodings"][0][len(kw["encodings"][0].split(",")[0]):]
        case["missions"] = rc.missions(**kw)
    # (drawn last, so that the seeds of earlier logs still name the same cases) UTF-16 units of every kind — surrogates alone, in pairs,
    # in rows — for the Missions that decode UTF-16: the wave path's slow mode and its way back
    if any(m["encoding"] in (2, 3) for m in case["missions"]) and r.random() < 0.5:   # (SX_ENC_UTF16LE / BE)
        from test_wave_core import utf16_soup
        be = r.random() < 0.5
        case["kind"] = "utf16soup"
        case["files"] = [utf16_soup(r, min(size, 300_000) // 2, be, r.choice([(50, 20, 8, 8, 6, 8), (40, 10, 14, 10, 10, 4), (70, 20, 2, 2, 5, 1)]))]
    # Round 6, SX_FUZZ_FUSED=1: the fused stage A (sx_fused.hip) — two or three of the UTF-8 / UTF-16LE / UTF-16BE Missions with filters whose
    # classifiers it holds, thresholds on both sides of the UTF-16 prefilter's 7, inputs with stretches of accepted units of every length at
    # every phase between bytes that pass or fail the prefilter; with and without the prefilter, fused and not
    if os.environ.get("SX_FUZZ_FUSED"):
        from test_gpu_parity import utf16_dense
        encs = r.sample(["utf-8", "utf-16le", "utf-16be"], r.choice([2, 3, 3]))
        if r.random() < 0.3:
            encs = [e + "," + r.choice(["", "3", "7", "9"]) + ",," + r.choice(["", "Greek", "Armenian", "Hebrew"]) for e in encs]
        if r.random() < 0.15:
            encs.insert(r.randrange(len(encs) + 1), r.choice(["koi8-r", "ascii", "utf-8,,,Cjk", "big5"]))
        kw = dict(encodings=encs, chars_min=r.choice(["4", "6", "7", "8", "10", "12", "20"]), output_line_len=r.choice([None, None, "10", "64", "100"]),
                  ascii_filter=r.choice([None, None, "All-Ctrl", "None"]), unicode_block_filter=r.choice([None, "African", "Common", "Cyrillic", "Arabic", "None"]),
                  grep_char=None, same_unicode_block=False, counter_offset=r.choice([None, "1000"]))
        case["kw"], case["missions"] = kw, rc.missions(**kw)
        k2 = r.choice(["utf16_dense", "utf16_dense", "synth", "soup", "random", "multi"])
        case["kind"] = "fused:" + k2
        if k2 == "utf16_dense": case["files"] = [utf16_dense(r, size)]
        elif k2 == "synth": case["files"] = [synth(r, size, 1 / 300)]
        elif k2 == "soup": case["files"] = [soup(r, min(size, 200_000))]
        elif k2 == "random": case["files"] = [r.randbytes(size)]
        else: case["files"] = [utf16_dense(r, r.randrange(1, 30000)) for _ in range(r.randrange(2, 5))]
        case["switches"] = r.choice([{}, {}, {}, {"SX_FUSED_PREFILTER": "0"}, {"SX_FUSED": "0"}, {"SX_PIECE_MIB": "0"}, {"SX_REGION_CAP": "2"}, {"SX_REGION_CAP": "0"},
                                     {"SX_WAVE_REPLAY": "0"}, {"SX_DEVICE_JOIN_MIN": "1"}])
        case["generic"] = False
    return case


def set_switches(sw):
    for k in ALL_SWITCHES:
        os.environ.pop(k, None)
    os.environ.update(sw)


def describe(c):
    return (f"kind={c['kind']} size={c['size']} chunk={c['chunk']} flush={c['flush']} sub={c['sub']} replay={c['replay']} "
            f"generic={c['generic']} switches={c['switches']}\n  flags={c['kw']}")
