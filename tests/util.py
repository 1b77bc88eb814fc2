"""Shared comparison helpers for the parity tests (tolerances are stated where they are used)."""
import torch


def max_abs_err(got, want):
    return (torch.as_tensor(got).double().cpu() - torch.as_tensor(want).double().cpu()).abs().max().item()


def assert_close(got, want, atol, rtol=0.0, name=""):
    """|got-want| <= atol + rtol*max|want| element-wise (north-star outputs: logits, loss, features,
    head gradients -- continuous functions of the inputs)."""
    got, want = torch.as_tensor(got).double().cpu(), torch.as_tensor(want).double().cpu()
    assert got.shape == want.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(want.shape)}"
    err = (got - want).abs().max().item()
    bound = atol + rtol * want.abs().max().item()
    assert err <= bound, f"{name}: max|d|={err:.3e} > {bound:.3e} (max|ref|={want.abs().max().item():.3e})"
    return err


def assert_close_robust(got, want, rel_l2, elem_tol, frac=0.97, name=""):
    """For encoder gradients / updated encoder weights.  fp32 re-association can flip a single ReLU or
    max-pool decision (probability ~ #elements * 1e-7), which moves one channel's gradient by a
    discrete amount; that is not an arithmetic error.  So: relative L2 error <= rel_l2 AND at least
    `frac` of the elements within elem_tol * max|want|."""
    got, want = torch.as_tensor(got).double().cpu().flatten(), torch.as_tensor(want).double().cpu().flatten()
    assert got.shape == want.shape, f"{name}: shape"
    ref = torch.linalg.norm(want).item()
    err = torch.linalg.norm(got - want).item()
    assert err <= rel_l2 * ref + 1e-12, f"{name}: relL2={err / max(ref, 1e-30):.3e} > {rel_l2:.1e}"
    within = ((got - want).abs() <= elem_tol * want.abs().max().item() + 1e-12).double().mean().item()
    assert within >= frac, f"{name}: only {within:.4f} of elements within {elem_tol:.1e}*max"
    return err / max(ref, 1e-30)
