"""Parser / comparator for the iteration tables the reference prints (`@info log_header` / `@info log_row`,
src/lm.jl:120-121,304; src/LevenbergMarquardt.jl:143-147) and keeps under benchmark/**/*.log -- the only end-to-end LM
traces of the reference that exist.  A log file holds several runs; each run is

    Info: FeasibilityResidual ... Problem name: <name>-feasres ...          (@info model,      lm.jl:28)
    Info: Parameters of the solver: facto = :LDL ... ite_max = 200          (lm.jl:29)
    Info: Tolerances: restol = ... rtol = ...                               (lm.jl:30)
    [ Info:   iter   f(x)   Δf   ‖Jᵀr‖   λ   ‖δ‖   ρ   status               (header)
    [ Info:      1   8.5e+05   0.0e+00   2.4e+07   4.2e+02   2.1e+00   1.0e+00   acc      (one row per iteration, 2 digits)
    Info: Generic Execution stats  status / objective value (full precision) / iterations / elapsed time   (lm.jl:416)

compare_trace() lines a device run (GenericExecutionStats.log of lm.py, same columns) up against a parsed run: same
accept/reject sequence, every printed number equal to the 2 significant digits the log keeps, final objective to full
precision.  The BAL files themselves are not in the image (no network), so the comparator is exercised on the
reference's logs against themselves and on synthetic runs; with the real `Data/` files it is the end-to-end parity check.
"""
import re
from dataclasses import dataclass, field

_NUM = r"[-+]?(?:\d+\.?\d*(?:[eE][-+]?\d+)?|NaN|Inf)"
_ROW = re.compile(r"^\[ Info:\s+(\d+)\s+(" + _NUM + r")\s+(" + _NUM + r")\s+(" + _NUM + r")\s+(" + _NUM + r")\s+(" + _NUM
                  + r")\s+(" + _NUM + r")\s+(\S+)\s*$")
_KV = re.compile(r"^[│└]\s+([^\s=:]+)\s*[=:]\s*(.+?)\s*$")
STATUS_TEXT = {  # SolverTools' descriptions of the status symbols of src/lm.jl:391-405
    "first-order stationary": "first_order", "solved to within acceptable tolerances": "acceptable",
    "stalled": "small_step", "small step": "small_step", "step too small": "small_step", "small residual": "small_residual",
    "maximum iteration": "max_iter", "maximum number of iterations": "max_iter", "unhandled exception": "exception",
    "unknown": "unknown",
}


@dataclass
class LoggedRun:
    problem: str = ""
    params: dict = field(default_factory=dict)       # facto, perm, normalize, linesearch, νd, νm, λ, ite_max, facto_type
    tolerances: dict = field(default_factory=dict)
    rows: list = field(default_factory=list)          # (iter, f, Δf, |J'r|, λ, |δ|, ρ, accepted: bool)
    status_text: str = ""
    status: str = ""
    objective: float = float("nan")
    dual_feas: float = float("nan")
    iterations: int = -1
    elapsed_time: float = float("nan")
    first_line: int = 0


def _val(s):
    s = s.strip()
    if s.startswith(":"):
        return s[1:]
    if s in ("true", "false"):
        return s == "true"
    try:
        return int(s)
    except ValueError:
        pass
    try:
        return float(s)
    except ValueError:
        return s


def parse_log(text):
    """-> list of LoggedRun, in file order.  `text`: the contents of one benchmark/**/*.log."""
    runs, cur, section = [], None, None
    for ln, line in enumerate(text.splitlines(), 1):
        line = line.rstrip("\n")
        if "Info: FeasibilityResidual" in line or (cur is None and "Problem name:" in line):
            cur = LoggedRun(first_line=ln)
            runs.append(cur)
            section = "model"
            continue
        if cur is None:
            continue
        if "Problem name:" in line:
            cur.problem = line.split("Problem name:")[1].strip()
            continue
        if "Info: Parameters of the solver" in line:
            section = "params"
            continue
        if "Info: Tolerances" in line:
            section = "tol"
            continue
        if "Info: Generic Execution stats" in line:
            section = "stats"
            continue
        m = _ROW.match(line)
        if m:
            it = int(m.group(1))
            vals = [float(m.group(k)) for k in range(2, 8)]
            cur.rows.append((it, *vals, m.group(8) in ("acc", "true")))
            section = "rows"
            continue
        if line.startswith("[ Info:") and "iter" in line and "f(x)" in line:
            section = "rows"
            continue
        if section == "stats":
            if line.lstrip("│└ ").startswith("status:"):
                cur.status_text = line.split("status:")[1].strip()
                cur.status = STATUS_TEXT.get(cur.status_text, cur.status_text)
            elif "objective value" in line:
                cur.objective = float(line.split(":")[1])
            elif "dual feasibility" in line:
                cur.dual_feas = float(line.split(":")[1])
            elif "iterations" in line:
                cur.iterations = int(line.split(":")[1])
            elif "elapsed time" in line:
                cur.elapsed_time = float(line.split(":")[1])
            continue
        m = _KV.match(line)
        if not m:
            continue
        key, val = m.group(1), m.group(2)
        if section == "params":
            cur.params[key] = _val(val)
        elif section == "tol":
            cur.tolerances[key] = _val(val)
    return runs


def _same_2digits(a, b):
    """a (full precision) prints as b (parsed from a %.1e field)?"""
    if a != a or b != b:
        return (a != a) and (b != b)
    return float("%.1e" % a) == b or abs(a - b) <= 0.051 * abs(b)  # one unit of the last printed digit of slack at .5 ties


def compare_trace(log_rows, ref, objective=None, rtol_objective=1e-6, max_rows=None):
    """Compare a run's log rows [(iter, f, Δf, |J'r|, λ, |δ|, ρ, accepted), ...] with a parsed LoggedRun.
    -> dict(ok, rows_compared, first_mismatch, acc_equal, objective_rel).  Every printed column must agree to the two
    digits the reference printed; the accept/reject sequence must be identical; `objective` (final, full precision) is
    compared with the log's "objective value" to rtol_objective (1e-6)."""
    n = min(len(log_rows), len(ref.rows)) if max_rows is None else min(len(log_rows), len(ref.rows), max_rows)
    first = None
    for k in range(n):
        a, b = log_rows[k], ref.rows[k]
        if int(a[0]) != b[0] or bool(a[7]) != b[7] or not all(_same_2digits(float(a[c]), b[c]) for c in range(1, 7)):
            first = dict(row=k, got=tuple(a), want=b)
            break
    acc_equal = [bool(r[7]) for r in log_rows[:n]] == [r[7] for r in ref.rows[:n]]
    out = dict(rows_compared=n, first_mismatch=first, acc_equal=acc_equal, same_length=len(log_rows) == len(ref.rows),
               objective_rel=None)
    ok = first is None and acc_equal and (max_rows is not None or out["same_length"])
    if objective is not None and ref.objective == ref.objective:
        out["objective_rel"] = abs(objective - ref.objective) / abs(ref.objective)
        ok = ok and out["objective_rel"] <= rtol_objective
    out["ok"] = ok
    return out
