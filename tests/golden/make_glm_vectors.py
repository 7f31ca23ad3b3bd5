"""Regenerates tests/golden/glm_vectors.json by running oracle/_ref/glm_probe, a probe compiled
against the glm headers the reference vendors (built by `make -C oracle ref`, only where
/root/reference exists).  The JSON holds IEEE-754 bit patterns of glm results for camera poses
and pseudo-random vectors; tests compare the oracle's and the product's arithmetic with them."""
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
PROBE = os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref", "glm_probe")

POSES = [  # pos, target, W, H
    ((6, 3, 12), (0, 0, 0), 800, 600),
    ((3, 4, 10), (0, 0, 0), 800, 600),
    ((-10, 0.5, -12), (5, 0, 5), 512, 512),
    ((0.3, 0.2, 3.1), (0.3, -5, 3.0), 512, 512),
    ((2, 9, 2), (2.001, 0, 2.002), 1920, 1080),
    ((0, 0, 0), (0, 0, -1), 1920, 1080),
]


def main():
    if not os.path.exists(PROBE):
        print("glm probe not built; nothing to do")
        return
    req = "".join("pose %r %r %r %r %r %r %d %d\n" % (*p, *t, w, h) for p, t, w, h in POSES) + "vec 777 64\n"
    out = subprocess.run([PROBE], input=req, capture_output=True, text=True, check=True).stdout.split("\n")
    lines = [l.split() for l in out if l]
    res = {"poses": [], "vec_seed": 777, "vec_n": 64, "vec": []}
    it = iter(lines)
    for p, t, w, h in POSES:
        rec = {"pos": p, "target": t, "W": w, "H": h}
        for key in ("cam", "inv", "p0", "p1", "p2"):
            l = next(it)
            assert l[0] == key
            rec[key] = l[1:]
        res["poses"].append(rec)
    for _ in range(64):
        rec = {}
        for key in ("dot", "cross", "norm", "min", "max", "div", "rdiv", "clamp"):
            l = next(it)
            assert l[0] == key
            rec[key] = l[1:]
        res["vec"].append(rec)
    with open(os.path.join(HERE, "glm_vectors.json"), "w") as f:
        json.dump(res, f, indent=0)
    print("wrote glm_vectors.json")


if __name__ == "__main__":
    main()
