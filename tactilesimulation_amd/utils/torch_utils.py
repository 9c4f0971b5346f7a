"""Host-side helpers the reference's environments use next to the simulator (utils/torch_utils.py), batched.

rotvec_mul composes rotation vectors (tactile_insertion_env.py uses it for the peg's orientation noise); here it takes
[..., 3] batches and is branch-free on the device.  tests/test_policy_and_utils.py pins it on golden vectors recorded from
the reference (tests/golden/policy_and_utils.npz, tools/make_policy_fixture.py).
"""
import torch


def rotvec_mul(a, b, eps=1e-7):
    """Rotation vector of R(a) R(b) for a, b [..., 3] (utils/torch_utils.py:18-39): quaternion product of the two half-angle
    quaternions, back to a rotation vector.  Like the reference: |a| < eps returns b, |b| < eps returns a, a composed
    angle < eps returns zeros."""
    an, bn = a.norm(dim=-1, keepdim=True), b.norm(dim=-1, keepdim=True)
    au, bu = a / an.clamp_min(1e-300), b / bn.clamp_min(1e-300)
    ca, sa, cb, sb = torch.cos(an / 2), torch.sin(an / 2), torch.cos(bn / 2), torch.sin(bn / 2)
    w = (ca * cb - ((au * sa) * (bu * sb)).sum(-1, keepdim=True)).clamp(-1.0, 1.0)
    cn = 2.0 * torch.arccos(w)
    v = ca * sb * bu + cb * sa * au + torch.linalg.cross(au * sa, bu * sb, dim=-1)
    c = cn * v / torch.sin(cn / 2).clamp_min(1e-300)
    c = torch.where(cn < eps, torch.zeros_like(c), c)
    c = torch.where(bn < eps, a, c)
    return torch.where(an < eps, b, c)
