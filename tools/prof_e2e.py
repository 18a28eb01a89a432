import cProfile, pstats, io, os, sys, time
sys.path.insert(0, os.getcwd())
sys.argv = ["x"]
os.environ["ZOT_TIMING"] = "0"
import importlib.util
spec = importlib.util.spec_from_file_location("be", "tools/bench_e2e.py")
src = open("tools/bench_e2e.py").read().split("R = int(float(sys.argv[1]))")[0]
exec(compile(src, "bench_e2e_head", "exec"))
R = 50_000_000
write_fastq("/tmp/a.fastq", R, 150)
from zotmer_amd import cli
cli.main_inner(["kmerize", "25", "/tmp/a.k25", "/tmp/a.fastq"])
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
cli.main_inner(["kmerize", "25", "/tmp/a.k25", "/tmp/a.fastq"])
pr.disable()
print("warm kmerize wall %.3f s" % (time.perf_counter() - t0))
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumtime").print_stats(45)
print(s.getvalue()[:9000])
os.remove("/tmp/a.fastq"); os.remove("/tmp/a.k25")
