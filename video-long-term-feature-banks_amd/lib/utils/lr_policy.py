"""Learning-rate schedules of cfg.SOLVER (same policies and warm-up rule as the reference's
lib/utils/lr_policy.py:41-157; scalar host logic, re-written table-driven)."""
import bisect

import numpy as np

from core.config import config as cfg


def get_step_index(cur_iter):
    """index of the STEPS interval that contains cur_iter (the last one is open-ended)"""
    steps = cfg.SOLVER.STEPS
    assert steps[0] == 0, "The first step should always start at 0."
    edges = list(steps) + [cfg.SOLVER.MAX_ITER]
    return min(bisect.bisect_right(edges, cur_iter) - 1, len(edges) - 2)


def _steps_with_lrs(it):
    return cfg.SOLVER.LRS[get_step_index(it)]


def _steps_with_relative_lrs(it):
    return cfg.SOLVER.LRS[get_step_index(it)] * cfg.SOLVER.BASE_LR


def _steps_with_decay(it):
    return cfg.SOLVER.BASE_LR * cfg.SOLVER.GAMMA ** get_step_index(it)


def _step(it):
    return cfg.SOLVER.BASE_LR * cfg.SOLVER.GAMMA ** (it // cfg.SOLVER.STEP_SIZE)


_POLICIES = {
    "steps_with_lrs": _steps_with_lrs,
    "steps_with_relative_lrs": _steps_with_relative_lrs,
    "steps_with_decay": _steps_with_decay,
    "step": _step,
}


def get_lr_func():
    try:
        return _POLICIES[cfg.SOLVER.LR_POLICY]
    except KeyError:
        raise NotImplementedError("Unknown LR policy: {}".format(cfg.SOLVER.LR_POLICY))


def get_lr_at_iter(it):
    """LR at iteration `it`, with the reference's gradual linear warm-up:
    for it in [0, WARMUP_END_ITER): linear from WARMUP_START_LR to policy(WARMUP_END_ITER)."""
    policy = get_lr_func()
    lr = np.float32(policy(it))
    warm = cfg.SOLVER.WARMUP
    if warm.WARMUP_ON and it < warm.WARMUP_END_ITER:
        lo = np.float32(warm.WARMUP_START_LR)
        hi = np.float32(policy(warm.WARMUP_END_ITER))
        lr = it * (hi - lo) / (warm.WARMUP_END_ITER - 1) + lo
    return np.float32(lr)
