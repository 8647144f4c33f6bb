"""bench.py's result line: the driver keeps the last 8 KB of stdout, so the ONE JSON line must fit there and still carry
the contract's keys, `roofline` and `cpu_baseline` (round-4 verdict: a 20 KB line was lost).  Canned run: the full result
dict of the round-4 bench (tests/golden/bench_full_r04.json)."""
import io
import json
import os
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def _full():
    with open(os.path.join(ROOT, "tests", "golden", "bench_full_r04.json")) as f:
        return json.load(f)


def test_result_line_fits_and_keeps_the_contract():
    full = _full()
    line = bench.short_line(full)
    text = json.dumps(line)
    assert len(text) < bench.MAX_LINE_BYTES <= 8192
    assert len(text) < 4096  # (headroom: a multi-GPU line adds latency_ms and a few checks)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "checks"):
        assert key in line, key
    assert line["value"] == round(full["value"], 6) and line["ms_per_step"] == round(full["ms_per_step"], 6)
    rf = line["roofline"]
    for key in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms",
                "launches_timed"):
        assert key in rf, key
    assert " " not in rf["kernel"] and len(rf["kernel"]) < 64  # a name, not prose
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    cb = line["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(cb) and "note" not in cb and "rust_probe" not in cb
    assert len(line["config"]["workload"]) <= 200 and "model" not in line["config"]
    assert all(not isinstance(v, (dict, list)) for v in line["extra"].values()) and len(line["extra"]) <= 32


def test_emit_puts_the_result_line_last_and_everything_else_before_it(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "EXTRA_FILE", str(tmp_path / "bench_extra.json"))
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.emit(_full())
    lines = buf.getvalue().splitlines()
    assert lines[-1].startswith("{") and json.loads(lines[-1])["roofline"]["bound"] == "hbm"
    assert len(lines[-1]) < 8192
    assert all(ln.startswith("extra: ") for ln in lines[:-1]) and len(lines) > 20
    assert json.load(open(tmp_path / "bench_extra.json"))["extra"]["batch256"]["ms_per_step"] > 0


def test_an_oversized_line_is_refused():
    full = _full()
    full["checks"] = {f"k{i}": "x" * 100 for i in range(100)}
    try:
        bench.short_line(full)
    except AssertionError as e:
        assert "result line" in str(e)
    else:
        raise AssertionError("an oversized line went through")
