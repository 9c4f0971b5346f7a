"""Bounding spheres of the contact pairs (csrc/tsim_hip.hip build_sched -> csrc/tsim_device.h ts_pair_bound): the formula the host uses,
restated in numpy on every shipped model — the device's pair cull is conservative iff every contact point lies inside its pair's sphere."""
import numpy as np

from tactilesimulation_amd.model.compiler import load_model
from tactilesimulation_amd.workloads import asset


def test_the_bounding_spheres_hold_every_contact_point():
    """What the host puts into the schedule (build_sched, tsim_hip.hip) restated in numpy on every shipped model: centre of the points' box, the
    largest distance to it; the device test is conservative iff every point is inside."""
    import tactilesimulation_amd.model.blob as BL
    for name in ("pusher", "dclaw_position_control", "tactile_insertion", "stable_grasp", "tactile_pad"):
        m = load_model(asset(name))
        I, F = np.asarray(m.I), np.asarray(m.F)
        ncpt, fc, op = I[BL.TSIM_IH_NCPT], I[BL.TSIM_IH_FOFF_CPT], I[BL.TSIM_IH_OFF_PAIR]
        P = np.stack([F[fc + a * ncpt: fc + (a + 1) * ncpt] for a in range(3)], 1)
        for pk in range(I[BL.TSIM_IH_NPAIR]):
            p0, n = I[op + pk * BL.TSIM_PI_SIZE + BL.TSIM_PI_PT0], I[op + pk * BL.TSIM_PI_SIZE + BL.TSIM_PI_NPT]
            pts = P[p0:p0 + n]
            c = (0.5 * (pts.min(0) + pts.max(0))).astype(np.float32).astype(np.float64)
            r = np.float32(np.linalg.norm(pts - c, axis=1).max() * (1 + 1e-6) + 1e-7)
            assert (np.linalg.norm(pts - c, axis=1) <= float(r)).all() and float(r) < 0.5
