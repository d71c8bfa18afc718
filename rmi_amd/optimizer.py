"""Pareto search over RMI configurations on ONE resident key array.

Mirror of rmi_lib/src/optimizer.rs: the same model lists and branching factors per
``RMI_OPTIMIZER_PROFILE`` (:14-58), the same two phases (:110-148), dominance rule (:173-187),
front (:60-73), narrowing (:75-107) and final ordering (:235-249); `train_for_size` is
train/mod.rs:128-154.  What changes is where the time goes: the keys stay in HBM, every
configuration is one sub-millisecond..millisecond device pass over them, and the only heavy host
work left is the exact, sequential root fit (rmi_hip_fit_root) -- computed once per
(root, branching factor), shared by all leaf types, and overlapped across configurations on a few
host threads (the reference's `par_iter` over whole trainings, optimizer.rs:220-231).

The one model of the lists the device path does not implement (lognormal, one of the `disk` profile's
extra tops: it needs libm's `ln` per key) is left out; `skipped_models()` names it.
"""
from __future__ import annotations

import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor, as_completed
from dataclasses import dataclass

from . import codegen, train

EPSILON = sys.float_info.epsilon

SUPPORTED_TOP = ("linear", "robust_linear", "linear_spline", "cubic", "radix", "radix8", "radix18", "radix22", "radix26", "radix28", "bradix", "normal", "loglinear")
SUPPORTED_LEAF = ("linear", "linear_spline", "cubic")


def _profile() -> str | None:
    p = os.environ.get("RMI_OPTIMIZER_PROFILE")
    if p is not None and p not in ("fast", "memory", "disk"):
        raise ValueError(f"Invalid optimizer profile {p}")                    # optimizer.rs:24, 38, 53
    return p


def _reference_top_only_layers() -> list[str]:                                # optimizer.rs:15-28
    p = _profile()
    if p == "fast":
        return ["robust_linear"]
    if p == "disk":
        return ["radix", "radix18", "radix22", "robust_linear", "normal", "lognormal", "loglinear"]
    return ["radix", "radix18", "radix22", "robust_linear"]


def _reference_anywhere_layers() -> list[str]:                                # optimizer.rs:30-41
    return ["linear", "cubic"] if _profile() == "fast" else ["linear", "cubic", "linear_spline"]


def top_only_layers() -> list[str]:
    return [m for m in _reference_top_only_layers() if m in SUPPORTED_TOP]


def anywhere_layers() -> list[str]:
    return [m for m in _reference_anywhere_layers() if m in SUPPORTED_LEAF]


def skipped_models() -> list[str]:
    return [m for m in _reference_top_only_layers() if m not in SUPPORTED_TOP]


def get_branching_factors() -> list[int]:                                     # optimizer.rs:43-58
    p = _profile()
    rng = range(6, 25, 2) if p == "fast" else (range(6, 28) if p == "disk" else range(6, 25))
    return [2 ** i for i in rng]


@dataclass
class RMIStatistics:                                                          # optimizer.rs:150-157
    models: str
    branching_factor: int
    average_log2_error: float
    max_log2_error: float
    size: int

    @staticmethod
    def from_trained(rmi: train.TrainedRMI) -> "RMIStatistics":               # :160-168
        return RMIStatistics(models=rmi.models, branching_factor=int(rmi.branching_factor),
                             average_log2_error=float(rmi.model_avg_log2_error),
                             max_log2_error=float(rmi.model_max_log2_error),
                             size=codegen.rmi_size(rmi.root.kind, rmi.leaf_kind, rmi.branching_factor, True,
                                                   0 if rmi.root.table is None else len(rmi.root.table)))

    def dominated_by(self, other: "RMIStatistics") -> bool:                   # :170-185
        if self.size < other.size:
            return False
        if self.average_log2_error < other.average_log2_error:
            return False
        if self.size == other.size and self.average_log2_error <= other.average_log2_error:
            return False
        log2_diff = abs(self.average_log2_error - other.average_log2_error)
        if self.size <= other.size and log2_diff < EPSILON:
            return False
        return True

    def has_config(self, models: str, branching_factor: int) -> bool:         # :187-189
        return self.models == models and self.branching_factor == branching_factor

    def to_grid_spec(self, namespace: str) -> dict:                           # :208-217
        return {"layers": self.models, "branching factor": self.branching_factor, "namespace": namespace,
                "size": self.size, "average log2 error": self.average_log2_error, "binary": True}


def pareto_front(results: list[RMIStatistics]) -> list[RMIStatistics]:       # optimizer.rs:60-73
    return [r for r in results if not any(r.dominated_by(v) for v in results)]


def narrow_front(results: list[RMIStatistics], desired_size: int) -> list[RMIStatistics]:   # optimizer.rs:75-107
    assert desired_size >= 2
    if len(results) <= desired_size:
        return list(results)
    tmp = sorted(results, key=lambda r: r.size)                               # stable, like sort_by
    best_mod = tmp.pop(0)
    while len(tmp) > desired_size - 1:
        # the two neighbours closest in size (first minimum, like min_by); drop the less accurate one
        gaps = [(tmp[i + 1].size / tmp[i].size, i) for i in range(len(tmp) - 1)]
        best = min(gaps, key=lambda g: g[0])
        i = best[1]
        if tmp[i].average_log2_error > tmp[i + 1].average_log2_error:
            tmp.pop(i)
        else:
            tmp.pop(i + 1)
    tmp.insert(0, best_mod)
    return tmp


def first_phase_configs() -> list[tuple[str, int]]:                           # optimizer.rs:110-126
    out = []
    for top in top_only_layers() + anywhere_layers():
        for bottom in anywhere_layers():
            for bf in get_branching_factors()[::5]:
                out.append((f"{top},{bottom}", bf))
    return out


def second_phase_configs(first_phase: list[RMIStatistics]) -> list[tuple[str, int]]:   # optimizer.rs:128-148
    qualifying = sorted({r.models for r in pareto_front(first_phase)})       # BTreeSet order
    out = []
    for models in qualifying:
        for bf in get_branching_factors():
            if any(v.has_config(models, bf) for v in first_phase):
                continue
            out.append((models, bf))
    return out


def measure_rmis(tr: train.Trainer, configs: list[tuple[str, int]], threads: int = 4,
                 root_cache: dict | None = None, progress=None, root_mode: str = "exact",
                 in_flight: int = 4) -> list[RMIStatistics]:
    """optimizer.rs:220-231.  Root fits (host, exact, sequential each) run `threads` at a time and are shared between the
    configurations that have the same (root, branching factor).  The leaf passes over the resident keys are ONE call of the
    library (rmi_hip_train_many: `in_flight` at a time, each on a context of its own that borrows the keys -- most
    configurations fill the GPU by themselves, but the ones with few, long leaves are a handful of sequential chains that
    leave it idle: in flight together they cost the time of one)."""
    root_cache = {} if root_cache is None else root_cache
    tr.download_keys()                           # host copy for the root fits
    parsed = [(train.parse_spec(m), m, bf) for m, bf in configs]
    need = []
    for (rk, _lk), _m, bf in parsed:
        if (rk, bf) not in root_cache and (rk, bf) not in need:
            need.append((rk, bf))
    # the leaf passes of the configurations whose roots are fitted go out in batches while the other root fits (a host core each,
    # about a second for 200 M keys) are still running
    done: list = [None] * len(parsed)
    pending = list(range(len(parsed)))
    errors: dict = {}

    def flush(final: bool) -> None:
        batch = [i for i in pending if (parsed[i][0][0], parsed[i][2]) in root_cache]
        if not batch or (not final and len(batch) < max(1, in_flight)):
            return
        res = tr.train_many([(root_cache[(parsed[i][0][0], parsed[i][2])], parsed[i][0][1], parsed[i][2]) for i in batch], in_flight=max(1, in_flight))
        for i, d in zip(batch, res):
            done[i] = d
            if d[0] != 0:
                # this configuration's own message (its position inside the batch is its number in the library's message)
                errors[i] = getattr(tr, "last_many_errors", {}).get(batch.index(i), getattr(tr, "last_many_error", ""))
        pending[:] = [i for i in pending if i not in batch]

    with ThreadPoolExecutor(max_workers=max(1, threads)) as root_pool:
        futs = {root_pool.submit(tr.fit_root, key[0], key[1], root_mode): key for key in need}
        for f in as_completed(futs):
            root_cache[futs[f]] = f.result()
            flush(False)
    flush(True)
    out = []
    for (rc, rmi), (_k, m, bf) in zip(done, parsed):
        if rc != 0:
            raise train.RMIError(rc, f"{m} {bf}: {errors.get(len(out), '')}")
        stats = RMIStatistics.from_trained(rmi)
        out.append(stats)
        if progress:
            progress(stats, rmi)
    return out


def find_pareto_efficient_configs(tr: train.Trainer, restrict: int, threads: int = 4,
                                  progress=None, root_mode: str = "exact", in_flight: int = 4) -> list[RMIStatistics]:     # optimizer.rs:233-249
    cache: dict = {}
    first = measure_rmis(tr, first_phase_configs(), threads, cache, progress, root_mode, in_flight)
    second = measure_rmis(tr, second_phase_configs(first), threads, cache, progress, root_mode, in_flight)
    front = pareto_front(second)                 # the reference takes the front of the second phase only
    front = narrow_front(front, restrict)
    front.sort(key=lambda r: r.average_log2_error)
    return front


def display_table(items: list[RMIStatistics], file=None) -> None:            # optimizer.rs:191-206
    file = file or sys.stdout
    rows = [("Models", "Branch", "   AvgLg2", "   MaxLg2", "   Size (b)")]
    for it in items:
        rows.append((it.models, f"{it.branching_factor:10}", f"     {it.average_log2_error:.5f}",
                     f"     {it.max_log2_error:.5f}", f"     {it.size}"))
    w = [max(len(r[c]) for r in rows) for c in range(5)]
    for r in rows:
        print(" ".join([r[0].ljust(w[0])] + [r[c].rjust(w[c]) for c in range(1, 5)]), file=file)


def train_for_size(tr: train.Trainer, max_size: int, threads: int = 4, root_mode: str = "exact") -> train.TrainedRMI:   # train/mod.rs:128-154
    t0 = time.perf_counter_ns()
    pareto = find_pareto_efficient_configs(tr, 1000, threads, root_mode=root_mode)
    fits = [c for c in pareto if c.size < max_size]
    if not fits:
        raise ValueError(f"Could not find any configurations smaller than {max_size}")
    cfg = fits[0]
    res = tr.train(cfg.models, cfg.branching_factor, root_mode=root_mode)
    res.build_time = time.perf_counter_ns() - t0
    return res
