"""Host-side profile (cProfile) of the bundled 121-frame sequence end to end; run on a GPU box: python tools/profile_sequence.py"""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import run_sequence
run_sequence.run("/tmp/seq_warm")           # warm-up (library load, first launches)
pr = cProfile.Profile()
pr.enable()
s, e, vo = run_sequence.run("/tmp/seq_prof")
pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(45)
print(out.getvalue()[:9000])
print(s)
