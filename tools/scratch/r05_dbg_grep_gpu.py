import sys, random, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_wave_core as t, refconfig as rc, stringsext_amd as sx, sxo_binding as sxo
flags=dict(encodings=["utf-16le"], chars_min="5", output_line_len="16", grep_char="101")
ms=rc.missions(**flags)
rng=random.Random(4003)
g=101
datas = [("grep text", t.grep_text(rng, 600_000, g)), ("text", t.text_lines(rng, 300_000)), ("long lines", t.text_lines(rng, 200_000, 100, 900)),
         ("random", rng.randbytes(300_000)), ("no grep", bytes(c for c in t.text_lines(rng, 200_000, 200, 2000) if c != g))]
name,data=datas[4]
data=data.decode('latin-1').encode('utf-16-le')
os.environ["SX_WAVE_REPLAY"]="1"; os.environ["SX_TIMING"]="1"
for b in ("1", None):
    if b: os.environ["SX_WAVE_BATCHES"]=b
    else: os.environ.pop("SX_WAVE_BATCHES",None)
    sc=sx.Scanner(ms, device=0)
    res=sc.scan(data, file_id=1); st=sc.stats()
    print("batches", b, "wave_windows", st.wave_windows, "repairs", st.wave_repairs, "findings", len(res), flush=True)
    res.free(); sc.close()
