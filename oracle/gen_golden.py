"""Regenerate tests/golden/ref_vectors.json from the REAL reference.

TEST INFRASTRUCTURE ONLY.  Builds the Boost-free subset of /root/reference with
`make -C oracle ref` (the reference's own sources, compiled where they lie, no stand-ins)
and stores the driver's output.  Run only where /root/reference exists; the committed
fixture is data (inputs and expected outputs), not reference source.
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def main():
    if not os.path.isdir("/root/reference/src"):
        sys.exit("/root/reference not present: golden vectors can only be regenerated next to the reference")
    subprocess.check_call(["make", "-C", HERE, "ref", "-j8"], stdout=subprocess.DEVNULL)
    out = subprocess.check_output([os.path.join(HERE, "_ref", "ref_driver")])
    data = json.loads(out)
    # reference outputs recorded in BASELINE.md section 2 (reference binary run by the surveyor)
    data["baseline_md_c1"] = {
        "command": "planning -D episodic-tiger -P po-uct -s 1024 --particle-amount 256 --runs 10000 --seed 1",
        "mean": "-2.64891", "var": "923.52", "count": "10000", "stder": "0.303895",
    }
    path = os.path.join(ROOT, "tests", "golden", "ref_vectors.json")
    with open(path, "w") as f:
        json.dump(data, f, separators=(",", ":"))
        f.write("\n")
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
