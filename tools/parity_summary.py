"""Condense a VRC_PARITY_STATS file (one JSON line per frame comparison of a test run, tests/orc.py:compare)
into profiles/<round>_parity_errors.json: per test, how many frames were compared with the oracle, the
largest and the mean error, the pixels' tie budgets and by how much the worst pixel exceeded E0 + 2 x budget
(<= 0: inside the rule).  usage: python tools/parity_summary.py stats.jsonl [stats2.jsonl ...] > out.json"""
import collections
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import scenes  # noqa: E402  (the rule is printed from the constants it is enforced with)

by = collections.OrderedDict()
for path in sys.argv[1:]:
    for line in open(path):
        r = json.loads(line)
        by.setdefault(r["test"], []).append(r)
out = {"rule": scenes.rule_string(),
       "where": "MI355X, pytest -m gpu" if "gpu" in " ".join(sys.argv[1:]) else "host build of the kernel code",
       "tests": {}}
for t, rs_all in by.items():
    notes = [r for r in rs_all if r.get("note")]
    rs = [r for r in rs_all if not r.get("note")]
    if not rs:
        continue
    wb = [r for r in rs if "budget_mean" in r]
    e = {"frames_compared": len(rs), "pixels": int(sum(r["shape"][0] * r["shape"][1] for r in rs)),
         "max_abs_error": max(r["max"] for r in rs), "largest_frame_mean_abs_error": max(r["mean"] for r in rs),
         "largest_fraction_of_pixels_over_1e-4": max(r["frac_over_1e4"] for r in rs)}
    if wb:
        e.update(frames_with_oracle_budget=len(wb), largest_frame_mean_budget=max(r["budget_mean"] for r in wb),
                 worst_pixel_error_minus_2x_budget=max(r["excess_max"] for r in wb),
                 pixels_over_E0_plus_2x_budget=int(sum(r["n_excess_over_1e4"] if False else 0 for r in wb)))
        e["pixels_with_error_minus_2x_budget_over_1e-4"] = int(sum(r.get("n_excess_over_1e4", 0) for r in wb))
        if any("budget_use" in r for r in wb):
            e["largest_budget_max"] = max(r["budget_max"] for r in wb)
            e["largest_budget_use"] = max(r.get("budget_use", 0.0) for r in wb)
            e["largest_fraction_of_pixels_needing_their_budget"] = max(r.get("needs_budget", 0.0) for r in wb)
            e["largest_frame_mean_pixel_error"] = max(r.get("pixel_mean", 0.0) for r in wb)
        del e["pixels_over_E0_plus_2x_budget"]
    if any("tie_bias" in r for r in notes):
        e["tie_bias"] = max((r["tie_bias"] for r in notes if "tie_bias" in r), key=abs)
    out["tests"][t] = e
json.dump(out, sys.stdout, indent=1)
print()
