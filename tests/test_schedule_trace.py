"""a12: the phase / part / early-exit sequence of optimize_envmap_ARMN.  The loop itself needs Mitsuba, so no reference
recording exists; the expected traces below are derived by hand from inverse_img_w_mi.py (line numbers in the comments)
for scripted loss sequences, and compared with what materialist_amd.schedule emits."""
import pytest

from materialist_amd.schedule import StepLR, TraceEvent, run_schedule


def decay_then_flat(start, rate, n):
    return lambda e: start * rate ** min(e, n)


def _ev(t):
    return (t.loop, t.phase, t.part, t.epoch, pytest.approx(t.lr, rel=1e-9), t.stop)


def test_steplr_only_steps_above_the_floor():
    # :363-365,431-432: StepLR(100, 0.8), stepped only while lr > 1.5e-4 -> 3e-4, 2.4e-4, 1.92e-4, 1.536e-4, 1.2288e-4, then frozen
    s = StepLR(3e-4)
    lrs = []
    for _ in range(600):
        lrs.append(s.lr)
        if s.lr > 1.5e-4:
            s.step()
    assert lrs[0] == 3e-4 and lrs[99] == 3e-4
    assert lrs[100] == pytest.approx(2.4e-4) and lrs[200] == pytest.approx(1.92e-4) and lrs[300] == pytest.approx(1.536e-4)
    assert lrs[400] == pytest.approx(1.2288e-4) and lrs[599] == pytest.approx(1.2288e-4)


def test_default_order_rm_a_opt_env_from_2():
    """BASELINE config 2: --opt_order 'rm a' --opt_env_from 2, opt_src 'arm'."""
    calls = {"env_end": [], "brdf_begin": [], "part_end": []}
    env_script = {1: lambda e: 0.1, 2: decay_then_flat(0.08, 0.98, 10), 3: lambda e: 0.07}
    brdf_script = {"rm": decay_then_flat(0.05, 0.99, 50), "a": lambda e: 0.04 * 0.997 ** e}
    tr = run_schedule(["rm", "a"], lambda l, e, lr: env_script[l](e), lambda l, p, e, lr: brdf_script[p](e), opt_src="arm", opt_env_from=2,
                      num_epochs=450, on_env_phase_end=lambda l, save: calls["env_end"].append((l, save)),
                      on_brdf_phase_begin=lambda l, which: calls["brdf_begin"].append((l, which)),
                      on_brdf_part_end=lambda l, p: calls["part_end"].append((l, p)))
    want = [
        # loop 1: one env epoch because loop_num < opt_env_from (:288-290); Adam 1e-3 in loop 1 (:226)
        (1, "env", "", 0, 1e-3, "loop<opt_env_from"),
        # 'rm': improves 1 %/epoch until epoch 50, then flat; patience 200//1, delta 0.1 % -> counter hits 200 at epoch 250;
        # StepLR stepped every epoch (lr > 1.5e-4): lr(250) = 3e-4 * 0.8^2
        (1, "brdf", "rm", 250, 3e-4 * 0.64, "early_stop"),
        (1, "brdf", "a", -1, 0.0, "skip 'a' in loop 1"),                        # :344-345
        # loop 2: Adam 1e-4, no scheduler (:229); 2 %/epoch until epoch 10, then flat; patience 100, delta 1 % -> stop at 110
        (2, "env", "", 110, 1e-4, "early_stop"),
        (2, "brdf", "rm", 150, 2.4e-4, "early_stop"),                           # patience 200//2 = 100 -> 50 + 100
        # 'a': -0.3 %/epoch never accumulates `patience` misses (every second epoch beats best*(1-0.5 %)) -> runs out of epochs;
        # lr frozen at 1.2288e-4 once it is <= 1.5e-4 (:431-432)
        (2, "brdf", "a", 449, 1.2288e-4, "num_epochs"),
        (3, "env", "", 100, 1e-4, "early_stop"),                                # flat from the start: stop after `patience` epochs
        (3, "end", "", -1, 0.0, "loop>=3"),                                     # :309-310
    ]
    assert [_ev(t) for t in tr] == want
    assert calls["env_end"] == [(1, False), (2, True), (3, True)]               # best_results only when loop >= opt_env_from (:302-303)
    assert calls["brdf_begin"] == [(1, "gt_or_ones"), (2, "optimized")]         # :317-327
    assert calls["part_end"] == [(1, "rm"), (2, "rm"), (2, "a")]


def test_global_early_stopping_ends_the_run():
    # :304-308: EarlyStopping(patience 2, 2.5 %) on the env-phase MSE of consecutive loops
    env_final = {1: 0.1, 2: 0.099, 3: 0.0985}
    tr = run_schedule(["arm"], lambda l, e, lr: env_final[l], lambda l, p, e, lr: 0.05, opt_src="arm", opt_env_from=0, num_epochs=150)
    assert [_ev(t) for t in tr] == [
        (1, "env", "", 100, 1e-3 * 0.8, "early_stop"),        # loop 1 uses StepLR(100, .8) on 1e-3 (:226-227): epoch 100 runs at 0.8e-3
        (1, "brdf", "arm", 149, 2.4e-4, "num_epochs"),        # flat loss, patience 200 > 150 epochs
        (2, "env", "", 100, 1e-4, "early_stop"),
        (2, "brdf", "arm", 100, 2.4e-4, "early_stop"),        # patience 100
        (3, "env", "", 100, 1e-4, "early_stop"),
        (3, "end", "", -1, 0.0, "early_stopping_all"),        # checked before `loop_num >= 3`
    ]


def test_rm_not_in_opt_src_and_skip():
    # :291-294: opt_src without 'rm' -> single env epoch in loop 1
    tr = run_schedule(["a"], lambda l, e, lr: 0.1 / l, lambda l, p, e, lr: 0.05, opt_src="a", opt_env_from=0, num_epochs=30)
    assert _ev(tr[0]) == (1, "env", "", 0, 1e-3, "rm not in opt_src")
    assert _ev(tr[1]) == (1, "brdf", "a", -1, 0.0, "skip 'a' in loop 1")
    assert tr[2].loop == 2 and tr[2].phase == "env" and tr[2].stop == "num_epochs"
    # :231-234,311-312: opt_src == 'skip' -> patience 500 and the run ends after the first env phase
    tr = run_schedule(["arm"], lambda l, e, lr: 0.1, lambda l, p, e, lr: 0.05, opt_src="skip", num_epochs=600)
    assert [_ev(t) for t in tr] == [(1, "env", "", 500, 1e-3 * 0.8 ** 5, "early_stop"), (1, "end", "", -1, 0.0, "skip")]
