"""Writes tests/golden/scotty_reference.json and scotty_mpc_expected.json: the ARRAYS (data only) of the reference's
test/scotty.json (the "Scotty dog" path its BicycleMPC fixture tracks, read by test/test_utils.cpp:240-295) and of
test/scotty_mpc.json (what bicycle_test.cpp:266-359 `TrackingMPC_2Solves` saved: solve_iters[200],
state_trajectory[201], input_trajectory[200], tracking_error[200]).

    python tests/golden/make_scotty_fixtures.py            (needs /root/reference; run in the build container)

Finding recorded in the fixture: scotty.json says N = 501 (the number of points) and tf = 50, so today's
bicycle_test.cpp:180 would take h = (float)(50 / 501); the saved run has tf = Nsim * h = 20.0 for Nsim = 200, i.e. it was
made with h = 0.1f (= 50 / 500, the spacing of the path itself: |dx| / v = 0.1).  The expected file carries h_used."""
import json
import os

REF = "/root/reference/test"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    ref = json.load(open(os.path.join(REF, "scotty.json")))
    mpc = json.load(open(os.path.join(REF, "scotty_mpc.json")))
    out_ref = {
        "source": "bjack205/altro test/scotty.json (data arrays only)",
        "N": ref["N"], "tf": ref["tf"],
        "state_trajectory": ref["state_trajectory"], "input_trajectory": ref["input_trajectory"],
    }
    out_mpc = {
        "source": "bjack205/altro test/scotty_mpc.json, written by test/bicycle_test.cpp:266-359 (data arrays only)",
        "N": mpc["N"], "tf": mpc["tf"], "h_used": "0.1f (tf / N of this file; see make_scotty_fixtures.py)",
        "solve_iters": mpc["solve_iters"], "state_trajectory": mpc["state_trajectory"],
        "input_trajectory": mpc["input_trajectory"], "tracking_error": mpc["tracking_error"],
    }
    for name, obj in (("scotty_reference.json", out_ref), ("scotty_mpc_expected.json", out_mpc)):
        with open(os.path.join(HERE, name), "w") as f:
            json.dump(obj, f, separators=(",", ":"))
        print("wrote", name, os.path.getsize(os.path.join(HERE, name)), "bytes")


if __name__ == "__main__":
    main()
