import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    # the GPU box has 256 host cores: torch's CPU oracle is much faster with a bounded thread count
    try:
        import torch
        torch.set_num_threads(min(16, os.cpu_count() or 1))
    except Exception:
        pass
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")


# ---- measured margins of the tolerance-gated asserts ------------------------------------------------------------
# `enable_assertion_pass_hook` (pytest.ini) hands every passing assert to pytest_assertion_pass with its source text
# and the evaluated explanation ("assert 3.1e-06 < 1e-05 ...").  Asserts that compare against a tolerance literal are
# kept -- worst measured value per (test, line) -- printed in the terminal summary and written to
# gpurun_out/measured_errors.json, so the margin of every parity gate is on record, not just pass / fail.
import json as _json
import re as _re

_TOL = _re.compile(r"<=?\s*\(?\s*(\d+(\.\d+)?e-\d+|tol\b|max\()")
_LEFT = _re.compile(r"^(?:assert\s+)?\(?([-+]?(\d+\.?\d*|\.\d+)([eE][-+]?\d+)?)\s*<")
_measured = {}
_NP = _re.compile(r"\b(?:np|numpy)\.float(?:16|32|64)\(([^()]*)\)")
_SAFE = _re.compile(r"^[-+*/() .0-9eE]+$")


def _left_value(first: str):
    """Measured (left) side of 'assert <left> < <tol>' from pytest's evaluated explanation: a number, possibly wrapped
    in np.float32(...), or plain arithmetic of numbers such as '(4.8e-07 / 0.146)' or 'abs((1.02 - 1.01))'."""
    text = _NP.sub(r"\1", first)
    if text.startswith("assert "):
        text = text[7:]
    left = _re.split(r"<=?", text, 1)[0].strip()
    m = _LEFT.match(left + " <")
    if m:
        return float(m.group(1))
    expr = left.replace("abs(", "(")
    if _SAFE.match(expr):
        try:
            return abs(float(eval(expr, {"__builtins__": {}}, {})))       # digits and operators only (checked above)
        except Exception:
            return None
    return None


def pytest_assertion_pass(item, lineno, orig, expl):
    if not _TOL.search(orig):
        return
    first = expl.strip().splitlines()[0][:200]
    val = _left_value(first)
    key = (item.nodeid, lineno)
    old = _measured.get(key)
    if old is None or (val is not None and (old["worst"] is None or val > old["worst"])):
        _measured[key] = {"test": item.nodeid, "line": lineno, "assert": " ".join(orig.split())[:160], "worst": val,
                          "evaluated": first, "hits": (old["hits"] if old else 0) + 1}
    else:
        old["hits"] += 1


def pytest_terminal_summary(terminalreporter):
    if not _measured:
        return
    rows = sorted(_measured.values(), key=lambda r: (r["test"], r["line"]))
    out_dir = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "measured_errors.json"), "w") as f:
            _json.dump(rows, f, indent=1)
    except OSError:
        pass
    tr = terminalreporter
    tr.section("measured values of the tolerance-gated asserts (worst per test line)")
    for r in rows:
        worst = "n/a" if r["worst"] is None else f"{r['worst']:.3g}"
        tr.write_line(f"{r['test'].split('/')[-1]}:{r['line']}  measured {worst}  |  {r['assert']}")
