import cProfile, pstats, sys, os, io
sys.argv = ["bench_fusion.py", "--cache_text", "--steps", "30", "--warmup", "5"]
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/tools")
pr = cProfile.Profile()
src = open(os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/tools/bench_fusion.py").read()
pr.enable()
exec(compile(src, "bench_fusion.py", "exec"), {"__name__": "__main__", "__file__": os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/tools/bench_fusion.py"})
pr.disable()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(28)
print(st.getvalue()[:6000])
