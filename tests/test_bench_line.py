"""bench.py prints ONE JSON line the driver parses: it must stay small (round 5's grew to 33 KB and BENCH_r05.json came back with
`parsed: null`) and carry the contract's keys, `roofline` and `cpu_baseline`.  Everything else belongs in bench_extras.json."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench          # noqa: E402
import bench_extras   # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config")


def recorded():
    """The round-5 record (33 KB as one line) as the `full` dict bench.py now keeps out of the line."""
    return json.load(open(os.path.join(ROOT, "profiles", "r05", "r05_zn_bench.json")))


def headline_like(full):
    r = full["roofline"]
    out = {k: full[k] for k in CONTRACT}
    out["config"] = {k: v for k, v in full["config"].items() if k != "ranks"}
    out["roofline"] = {k: r[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "frac_physical",
                                         "bytes_per_launch", "avg_launch_us", "steps_per_launch", "bytes_per_step", "avg_alive_tracks")}
    c = full["cpu_baseline"]
    out["cpu_baseline"] = {"value": c["value"], "unit": c["unit"], "cores": c["cores"], "kind": c["kind"], "host_cores": c["host_cores"],
                           "sample": c["sample"][:160], "value_8_threads": c["port_8_threads"]["value"]}
    out["parity"] = full["parity"]
    out["kernels"] = {"finalize_avg_us": full["kernels"]["finalize_avg_us"]}
    out["extras"] = bench_extras.summary(full)
    out["extras_file"] = "bench_extras.json"
    return out


def test_line_is_small_and_complete():
    out = headline_like(recorded())
    line = bench.compact_line(out)
    assert "\n" not in line and len(line) < bench.LINE_LIMIT <= 4096
    d = json.loads(line)
    for k in CONTRACT + ("roofline", "cpu_baseline", "parity", "extras"):
        assert k in d, k
    assert d["vs_baseline"] is None and d["config"]["workload"].startswith("configs[1]")
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    # the figures the reviews quote survive in the summary
    assert d["extras"]["1080p_opt_realistic"]["ms"] > 0 and d["extras"]["davis_x16"]["ms"] > 0


def test_optional_parts_are_dropped_before_the_line_overflows():
    out = headline_like(recorded())
    out["extras"] = {"pad%d" % i: "x" * 100 for i in range(60)}       # 6 KB of extras
    d = json.loads(bench.compact_line(out))
    assert "extras" not in d and "roofline" in d and "cpu_baseline" in d
    out["config"]["workload"] = "w" * 5000                               # nothing optional left to drop: refuse, never print a long line
    try:
        bench.compact_line(out)
    except ValueError:
        pass
    else:
        raise AssertionError("an over-long line was serialised")


def test_summary_reports_failed_and_skipped_figures():
    s = bench_extras.summary({"secondary_hard": {"error": "RuntimeError: boom"}, "secondary": {"skipped": "extras budget"}})
    assert "error" in s["1080p_opt_hard"] and "error" in s["sintel_opt"]


def test_main_prints_the_line_last():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.count("print(compact_line(out), flush=True)") == 1      # the only print of main(): the last stdout line is the JSON
    assert "print(json.dumps(" not in src
