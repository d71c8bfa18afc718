"""Host logic of the Pareto search (rmi_amd/optimizer.py) against the rules of
rmi_lib/src/optimizer.rs: configuration lists, dominance, front, narrowing."""
import pytest

from rmi_amd import optimizer as opt

S = opt.RMIStatistics


def test_branching_factors_and_lists(monkeypatch):
    monkeypatch.delenv("RMI_OPTIMIZER_PROFILE", raising=False)
    assert opt.get_branching_factors() == [2 ** i for i in range(6, 25)]            # optimizer.rs:45
    assert opt._reference_top_only_layers() == ["radix", "radix18", "radix22", "robust_linear"]
    assert opt.anywhere_layers() == ["linear", "cubic", "linear_spline"]
    assert opt.skipped_models() == [] and opt.top_only_layers() == opt._reference_top_only_layers()
    # (tops + anywhere) x anywhere x every 5th branching factor (6, 11, 16, 21)
    cfgs = opt.first_phase_configs()
    assert len(cfgs) == (4 + 3) * 3 * 4
    assert cfgs[0] == ("radix,linear", 64) and cfgs[1] == ("radix,linear", 2 ** 11)
    monkeypatch.setenv("RMI_OPTIMIZER_PROFILE", "fast")
    assert opt.get_branching_factors() == [2 ** i for i in range(6, 25, 2)]
    assert opt.top_only_layers() == ["robust_linear"] and opt.anywhere_layers() == ["linear", "cubic"]
    assert len(opt.first_phase_configs()) == 3 * 2 * 2 and opt.skipped_models() == []
    monkeypatch.setenv("RMI_OPTIMIZER_PROFILE", "disk")
    assert opt.get_branching_factors()[-1] == 2 ** 27
    assert opt.skipped_models() == ["lognormal"]
    monkeypatch.setenv("RMI_OPTIMIZER_PROFILE", "bogus")
    with pytest.raises(ValueError):
        opt.get_branching_factors()


def test_dominance_rule():
    a = S("m", 64, 5.0, 9.0, 1000)
    assert not a.dominated_by(S("m", 64, 6.0, 9.0, 2000))       # smaller than the other
    assert not a.dominated_by(S("m", 64, 6.0, 9.0, 500))        # more accurate than the other
    assert a.dominated_by(S("m", 64, 4.0, 9.0, 500))            # other is smaller and more accurate
    assert a.dominated_by(S("m", 64, 5.0, 9.0, 500))            # same error, other smaller: dominated...
    assert not a.dominated_by(S("m", 64, 5.0, 9.0, 1000))       # ...unless the sizes are equal too
    assert a.dominated_by(S("m", 64, 4.0, 9.0, 1000))           # same size, other more accurate
    assert not S("m", 64, 4.0, 9.0, 1000).dominated_by(a)
    assert a.dominated_by(S("m", 64, 4.0, 9.0, 999))


def test_front_and_narrowing():
    pts = [S("a", 64, 10.0, 0, 100), S("b", 64, 8.0, 0, 200), S("c", 64, 9.0, 0, 300),   # c dominated by b
           S("d", 64, 5.0, 0, 400), S("e", 64, 4.9, 0, 410), S("f", 64, 2.0, 0, 5000)]
    front = opt.pareto_front(pts)
    assert [p.models for p in front] == ["a", "b", "d", "e", "f"]
    assert opt.narrow_front(front, 10) == front
    # narrowing to 4: the smallest stays; closest sizes are d/e (ratio 1.025): drop the less accurate d
    nf = opt.narrow_front(front, 4)
    assert [p.models for p in nf] == ["a", "b", "e", "f"]
    # to 3: of (b,e) ratio 2.05 and (e,f) ratio 12.2 -> pair (b,e): drop b (less accurate)
    assert [p.models for p in opt.narrow_front(front, 3)] == ["a", "e", "f"]
    with pytest.raises(AssertionError):
        opt.narrow_front(front, 1)


def test_second_phase_skips_measured_configs(monkeypatch):
    monkeypatch.setenv("RMI_OPTIMIZER_PROFILE", "fast")
    first = [S("linear,linear", 64, 9.0, 0, 1000), S("linear,linear", 2 ** 16, 3.0, 0, 10 ** 6),
             S("cubic,linear", 64, 9.5, 0, 1016), S("cubic,linear", 2 ** 16, 3.5, 0, 10 ** 6 + 16)]
    second = opt.second_phase_configs(first)
    assert all(m == "linear,linear" for m, _ in second)          # cubic,linear is dominated everywhere
    assert [bf for _, bf in second] == [2 ** i for i in range(8, 25, 2) if i != 16]
