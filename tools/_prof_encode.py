import cProfile, pstats, os, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
import encode_bench as eb
os.environ["MFAR_ENCODE_AUTOCAST"] = "bf16"
# reuse run() but profile only the encodes: monkeypatch time-critical part by profiling the whole run with small docs
pr = cProfile.Profile()
pr.enable()
res = eb.run(20000, 64, sweep=False, quiet=True)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
print(res["fp32"], res["autocast_bf16"])
