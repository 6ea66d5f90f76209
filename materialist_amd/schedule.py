"""The alternating env / BRDF optimisation schedule of `optimize_envmap_ARMN` (a12 of SURVEY.md section 8a), as a
driver that is independent of what an iteration computes.

Restated from inverse_img_w_mi.py:
  outer loop            :223-224,304-312   `while loop_num <= 10`, effective end at loop 3, global EarlyStopping(2, 2.5 %)
  env phase             :225-294           Adam 1e-3 + StepLR(100, .8) in loop 1, Adam 1e-4 afterwards; EarlyStopping(100 | 500, 1 %);
                                           single epoch when loop < opt_env_from, or when 'rm' not in opt_src in loop 1
  BRDF phase            :343-345,359-365,425-432,471-477
                                           parts in `optimize_order`, part 'a' skipped in loop 1; lr 3e-4, StepLR(100, .8) stepped
                                           only while lr > 1.5e-4; EarlyStopping(200 // loop, 0.5 % if 'a' in part else 0.1 %)
  light of the BRDF phase :317-327         ones / ground-truth envmap in loop 1 when loop < opt_env_from, else the best envmap

The callbacks run the actual iterations (`env_step(loop, epoch, lr) -> loss_mse`, `brdf_step(loop, part, epoch, lr) ->
loss_mse`); every decision is appended to `trace` so that tests can compare the emitted sequence with one derived by
hand from the cited lines.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

from .loop import EarlyStopping

NUM_EPOCHS = 5000          # :211
ENV_LR_FIRST, ENV_LR_LATER = 1e-3, 1e-4   # :226,229
BRDF_LR = 3e-4             # :359,470
LR_STEP, LR_GAMMA, LR_FLOOR = 100, 0.8, 1.5e-4   # :227,363,431


@dataclass
class TraceEvent:
    loop: int
    phase: str        # 'env' | 'brdf' | 'end'
    part: str         # '' for env
    epoch: int        # last epoch index that ran in this phase/part (-1 for 'end')
    lr: float         # learning rate of that last epoch
    stop: str         # why the phase / run ended


class StepLR:
    """torch.optim.lr_scheduler.StepLR on a scalar: lr = base * gamma ** (calls // step_size)."""

    def __init__(self, base_lr: float, step_size: int = LR_STEP, gamma: float = LR_GAMMA):
        self.base, self.step_size, self.gamma, self.calls = base_lr, step_size, gamma, 0

    @property
    def lr(self) -> float:
        return self.base * self.gamma ** (self.calls // self.step_size)

    def step(self) -> None:
        self.calls += 1


def brdf_patience(loop_num: int) -> int:
    return 200 // loop_num     # :361,473


def brdf_min_delta(part: str) -> float:
    return 0.005 if "a" in part else 0.001   # :360-363,473-476


def run_schedule(optimize_order: Sequence[str], env_step: Callable[[int, int, float], float],
                 brdf_step: Callable[[int, str, int, float], float], opt_src: str = "arm", opt_env_from: int = 0,
                 num_epochs: int = NUM_EPOCHS, on_env_phase_end: Optional[Callable[[int, bool], None]] = None,
                 on_brdf_phase_begin: Optional[Callable[[int, str], None]] = None,
                 on_brdf_part_begin: Optional[Callable[[int, str], None]] = None,
                 on_brdf_part_end: Optional[Callable[[int, str], None]] = None, trace: Optional[List[TraceEvent]] = None,
                 brdf_part_runner: Optional[Callable[[int, str, int, float, int], tuple]] = None,
                 env_phase_runner: Optional[Callable[[int, Callable[[int], float], int, float, int], tuple]] = None) -> List[TraceEvent]:
    """`brdf_part_runner(loop, part, patience, min_delta, num_epochs) -> (last_epoch, last_lr, stop)` and
    `env_phase_runner(loop, lr_of_epoch, patience, min_delta, max_epochs) -> (last_epoch, stop, last_loss_mse)` may replace the
    per-epoch loops with ones that keep the EarlyStopping state machine on the device (FusedBrdfPhase / FusedEnvPhase);
    `max_epochs` is 1 where the reference breaks after the first epoch."""
    trace = [] if trace is None else trace
    early_stopping_all = EarlyStopping(patience=2, min_delta=0.025)                    # :222
    loop_num = 0
    while loop_num <= 10:                                                              # :223
        loop_num += 1
        # ------------------------------------------------------------------ env phase
        sched = StepLR(ENV_LR_FIRST) if loop_num == 1 else None                        # :225-229
        patience_env = 500 if opt_src == "skip" else 100                               # :231-234
        early_stopping = EarlyStopping(patience=patience_env, min_delta=0.01)          # :235
        loss_mse, stop, epoch, lr = float("nan"), "num_epochs", -1, ENV_LR_LATER
        single = "loop<opt_env_from" if loop_num < opt_env_from else (
            "rm not in opt_src" if ("rm" not in opt_src and loop_num == 1 and opt_src != "skip") else "")
        if env_phase_runner is not None:
            lr_of = (lambda e: ENV_LR_FIRST * LR_GAMMA ** (e // LR_STEP)) if loop_num == 1 else (lambda e: ENV_LR_LATER)
            epoch, stop, loss_mse = env_phase_runner(loop_num, lr_of, patience_env, 0.01, 1 if single else num_epochs)
            lr = lr_of(max(epoch, 0))
            if stop != "early_stop" and single:
                stop = single
        for epoch in (range(num_epochs) if env_phase_runner is None else ()):
            lr = sched.lr if sched is not None else ENV_LR_LATER
            loss_mse = env_step(loop_num, epoch, lr)
            early_stopping(loss_mse)                                                   # :250
            if sched is not None:
                sched.step()                                                           # :253-254
            if early_stopping.early_stop:                                              # :285-287
                stop = "early_stop"
                break
            if loop_num < opt_env_from:                                                # :288-290
                stop = "loop<opt_env_from"
                break
            if "rm" not in opt_src and loop_num == 1 and opt_src != "skip":            # :291-294
                stop = "rm not in opt_src"
                break
        trace.append(TraceEvent(loop_num, "env", "", epoch, lr, stop))
        if on_env_phase_end is not None:
            on_env_phase_end(loop_num, loop_num >= opt_env_from)                       # save best_results iff loop >= opt_env_from (:302-303)
        early_stopping_all(loss_mse)                                                   # :304
        if early_stopping_all.early_stop:                                              # :305-308
            trace.append(TraceEvent(loop_num, "end", "", -1, 0.0, "early_stopping_all"))
            break
        if loop_num >= 3:                                                              # :309-310
            trace.append(TraceEvent(loop_num, "end", "", -1, 0.0, "loop>=3"))
            break
        if opt_src == "skip":                                                          # :311-312
            trace.append(TraceEvent(loop_num, "end", "", -1, 0.0, "skip"))
            break
        # ------------------------------------------------------------------ BRDF phase
        if on_brdf_phase_begin is not None:                                            # :317-327
            on_brdf_phase_begin(loop_num, "gt_or_ones" if (loop_num < opt_env_from and loop_num == 1) else "optimized")
        for part in optimize_order:
            if part == "a" and loop_num <= 1:                                          # :344-345
                trace.append(TraceEvent(loop_num, "brdf", part, -1, 0.0, "skip 'a' in loop 1"))
                continue
            if on_brdf_part_begin is not None:
                on_brdf_part_begin(loop_num, part)
            if brdf_part_runner is not None:
                epoch, lr, stop = brdf_part_runner(loop_num, part, brdf_patience(loop_num), brdf_min_delta(part), num_epochs)
                trace.append(TraceEvent(loop_num, "brdf", part, epoch, lr, stop))
                if on_brdf_part_end is not None:
                    on_brdf_part_end(loop_num, part)
                continue
            sched = StepLR(BRDF_LR)
            early_stopping = EarlyStopping(patience=brdf_patience(loop_num), min_delta=brdf_min_delta(part))
            stop, epoch, lr = "num_epochs", -1, BRDF_LR
            for epoch in range(num_epochs):
                lr = sched.lr
                loss = brdf_step(loop_num, part, epoch, lr)
                early_stopping(loss)                                                   # :428,550
                if lr > LR_FLOOR:                                                      # :431-432,553-554
                    sched.step()
                if early_stopping.early_stop:                                          # :448-458
                    stop = "early_stop"
                    break
            trace.append(TraceEvent(loop_num, "brdf", part, epoch, lr, stop))
            if on_brdf_part_end is not None:
                on_brdf_part_end(loop_num, part)                                       # restore best maps, save_results (:460-465,583-590)
    return trace
