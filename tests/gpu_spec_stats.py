"""Experiment: counters of the parse-following matcher k_follow (needs zlib_amd/libzamd_stats.so: make zlib_amd/libzamd_stats.so)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ZAMD_MATCH_FOLLOW"] = "1"
import zlib_amd
zlib_amd.LIB_PATH = os.path.join(os.path.dirname(zlib_amd.__file__), "libzamd_stats.so")
lib = zlib_amd.load_library()
n = 256 << 20
names = ["cand_steps", "lane_slots", "searches", "hops", "service_passes(wave) / spins", "walk_cycles(wave)", "full_compares / waiting lane-rounds", "total_cycles(wave)"]
for kind in (0, 2, 3, 4):
    data = zlib_amd.corpus(n, kind, 5, threads=8)
    eng = zlib_amd.Engine(0, workspace_bytes=16 << 30)
    out = (ctypes.c_ulonglong * 8)()
    lib.zamd_debug_spec_stats(out, 1)
    stream, offs = eng.deflate_chunks(data, level=6)[:2]
    lib.zamd_debug_spec_stats(out, 1)
    v = list(out)
    print("kind", kind, "ratio %.3f" % (len(stream) / n))
    for k, x in zip(names, v):
        print("   %-24s %14d  %.3f /byte" % (k, x, x / n))
    print("   utilisation of walk slots %.3f; walk share of cycles %.3f; cand per search %.1f; searches per hop %.2f" % (
        v[0] / max(v[1], 1), v[5] / max(v[7], 1), v[0] / max(v[2], 1), v[2] / max(v[3], 1)))
    eng.close()
