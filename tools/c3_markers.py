"""bench.py --only c3 with another marker count (an experiment: is configs[2]'s tail slower because a 100,000-marker
bit-plane row is 25,008 bytes, not a whole number of 128-byte lines?).  usage: python tools/c3_markers.py 100352 [bench args]"""
import sys

M = sys.argv[1]
src = open("bench.py").read()
old = "N, M, E, n = 2048, 100_000, 4, 2048"
assert old in src
sys.argv = ["bench.py", "--only", "c3"] + sys.argv[2:]
sys.path.insert(0, ".")
exec(compile(src.replace(old, f"N, M, E, n = 2048, {M}, 4, 2048"), "bench.py", "exec"), {"__name__": "__main__", "__file__": "bench.py"})
