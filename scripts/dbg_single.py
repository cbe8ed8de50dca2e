import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen as Dg
import triple_accel_amd as T
g = Dg.rng(3)
x = Dg.rand_str(g, 256); y = Dg.mutate(g, x, 12)
os.environ["TA_DEBUG"] = "1"
print("levenshtein", T.levenshtein(x, y), T.last_launch_info()["kernel"])
print("rdamerau", T.rdamerau(x, y), T.last_launch_info()["kernel"])
print("simd_k 32", T.levenshtein_simd_k(x, y, 32), T.last_launch_info()["kernel"])
print("exp", T.levenshtein_exp(x, y), T.last_launch_info()["kernel"])
