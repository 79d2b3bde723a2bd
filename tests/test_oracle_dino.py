"""Pin the DINO part of the CPU oracle (oracle/ref_ops.py dino_head / dino_loss / dino_center_update) against outputs of
the reference's vit.DINOHead and loss.DINOLoss (golden G8, tools/gen_goldens.py dino) -- SURVEY section 8 row F2."""
import torch

from golden_util import Golden
from oracle import ref_ops as R
from oracle.formula import check_summary, fill, fill_state_dict, name_seed


def head_params():
    shapes = {"mlp.0.weight": (2048, 384), "mlp.0.bias": (2048,), "mlp.2.weight": (2048, 2048), "mlp.2.bias": (2048,),
              "mlp.4.weight": (256, 2048), "mlp.4.bias": (256,), "last.weight_g": (4096, 1), "last.weight_v": (4096, 256)}
    sd = fill_state_dict({k: torch.zeros(s) for k, s in shapes.items()})
    sd["last.weight_g"] = fill((4096, 1), name_seed("last.weight_g"), 0.3, 1.0)
    return {k: v.double().requires_grad_(True) for k, v in sd.items()}


def test_dino_head_fp64():
    g = Golden("g8_dino")
    P = head_params()
    assert [str(n) for n in g.arr("head.param_names")] == list(P.keys())
    x = fill((6, 384), 81, 1.0).double().requires_grad_(True)
    out = R.dino_head(x, P)
    check_summary(out, g.rec("head.out"), 5e-7, "dino head out")
    grads = torch.autograd.grad((out * fill(out.shape, 82, 1.0).double()).sum(), [x] + list(P.values()))
    check_summary(grads[0], g.rec("head.dx"), 5e-7, "dino head dx")
    for n, gr in zip(P.keys(), grads[1:]):
        check_summary(gr, g.rec(f"head.grad.{n}"), 5e-7, n)


def test_dino_loss_fp64():
    g = Golden("g8_dino")
    center = fill((1, 4096), 83, 0.2).double()
    student = fill((12, 4096), 84, 2.0).double().requires_grad_(True)
    teacher = fill((6, 4096), 85, 2.0).double()
    tt = float(g.arr("loss.teacher_temp"))
    assert abs(tt - (0.04 + (0.07 - 0.04) * 5 / 29)) < 1e-8             # fp32 linspace(0.04, 0.07, 30)[5]  (loss.py:108-118)
    loss = R.dino_loss(student, teacher, center, 4, 0.1, tt)
    assert abs(loss.item() - float(g.arr("loss.value"))) <= 1e-9 * abs(float(g.arr("loss.value")))
    (ds,) = torch.autograd.grad(loss, [student])
    check_summary(ds, g.rec("loss.dstudent"), 5e-7, "dino loss d student")
    check_summary(R.dino_center_update(center, teacher, 0.9), g.rec("loss.center_after"), 5e-7, "dino center")
