"""rulgnn_adam_bn_step_f32 (include/rulgnn.h): the optimizer step and the BatchNorm running-statistics update behind a data-parallel
all-reduce in one launch -- bit-identical to rulgnn_adam_step[_guarded]_f32 followed by rulgnn_bn_running_update[_guarded]_f32
(reference: optimizer.step() in algorithms/algorithms.py:474-478 and nn.BatchNorm1d's running statistics, models/ST_GCN/Model.py)."""
import ctypes as C

import pytest


def _state(n, L, seed, dev):
    import torch
    g = torch.Generator().manual_seed(seed)
    mk = lambda *s: torch.randn(*s, generator=g)
    k = max(n, 1)                              # n = 0: valid pointers, no element
    p, gr, m = mk(k), mk(k), 0.1 * mk(k)
    v = (0.1 * mk(k)) ** 2
    bn = mk(L * 2 * 2 * 10)
    batch = mk(L * 2 * 2 * 10).abs() + 0.5       # moments: E[z^2] >= E[z]^2 not required, the kernel clamps at zero
    return [t.to(dev) for t in (p, gr, m, v, bn, batch)]


@pytest.mark.gpu
@pytest.mark.parametrize("n,L", [(1083, 2), (7, 1), (300000, 8), (0, 2)])
@pytest.mark.parametrize("from_moments", [0, 1])
@pytest.mark.parametrize("guard", [None, 0.25, float("nan")])
def test_one_launch_equals_the_two_kernels(n, L, from_moments, guard):
    import torch
    from gnn_rul_benchmarking_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    hp = (3, 1e-3, 0.9, 0.999, 1e-8, 1e-4, 1.0)          # step, lr, betas, eps, weight decay, gradient scale
    count, mom = 1400, 0.1
    a = _state(n, L, 5, dev)
    b = [t.clone() for t in a]
    gd = None if guard is None else torch.tensor([guard], device=dev)
    gp = None if gd is None else gd.data_ptr()
    # the two kernels
    p, gr, m, v, bn, batch = a
    if n > 0:
        if gd is None:
            _lib.check(lib.rulgnn_adam_step_f32(p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), n, *hp, st), "adam")
        else:
            _lib.check(lib.rulgnn_adam_step_guarded_f32(p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), n, *hp, gp, st), "adam")
    if gd is None:
        _lib.check(lib.rulgnn_bn_running_update_f32(bn.data_ptr(), batch.data_ptr(), L, count, mom, from_moments, st), "bn")
    else:
        _lib.check(lib.rulgnn_bn_running_update_guarded_f32(bn.data_ptr(), batch.data_ptr(), L, count, mom, from_moments, gp, st), "bn")
    # one launch
    p2, gr2, m2, v2, bn2, batch2 = b
    _lib.check(lib.rulgnn_adam_bn_step_f32(p2.data_ptr(), gr2.data_ptr(), m2.data_ptr(), v2.data_ptr(), n, *hp, bn2.data_ptr(),
                                           batch2.data_ptr(), L, count, mom, from_moments, gp, st), "adam_bn")
    torch.cuda.synchronize()
    for x, y, name in zip(a, b, ("params", "grads", "exp_avg", "exp_avg_sq", "bn", "batch")):
        assert torch.equal(x, y), name
    if guard is not None and guard != guard:            # rejected step: nothing moved
        for x, y in zip(a, _state(n, L, 5, dev)):
            assert torch.equal(x, y)
    elif n > 0:
        assert not torch.equal(a[0], _state(n, L, 5, dev)[0])


@pytest.mark.gpu
def test_rejects_bad_arguments():
    import torch
    from gnn_rul_benchmarking_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    p, gr, m, v, bn, batch = _state(16, 2, 1, dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ok = lambda **k: lib.rulgnn_adam_bn_step_f32(
        k.get("p", p.data_ptr()), gr.data_ptr(), m.data_ptr(), v.data_ptr(), 16, k.get("step", 1), 1e-3, 0.9, 0.999, 1e-8, 0.0, 1.0,
        k.get("bn", bn.data_ptr()), batch.data_ptr(), k.get("L", 2), k.get("count", 10), 0.1, 0, None, st)
    assert ok() == _lib.OK
    assert ok(step=0) == _lib.EINVAL
    assert ok(L=0) == _lib.EINVAL and ok(L=9) == _lib.EINVAL
    assert ok(count=0) == _lib.EINVAL
    assert ok(p=None) != _lib.OK and ok(bn=None) != _lib.OK
    torch.cuda.synchronize()
