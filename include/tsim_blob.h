/* tsim_blob.h — flat model-blob layout (DATA FORMAT ONLY, no algorithm).
 *
 * A model (one XML of the reference, e.g. envs/assets/pusher/pusher.xml) is compiled on the host
 * by tactilesimulation_amd/model/compiler.py into two flat arrays:
 *     int32   I[]   topology, indices, counts
 *     float64 F[]   every numeric parameter (converted to the kernel's real type on upload)
 * Both the HIP kernels (tactilesimulation_amd/csrc) and the CPU oracle (oracle/) read this layout;
 * it is mirrored field-for-field in tactilesimulation_amd/model/blob.py.
 *
 * Conventions
 *   - link 0 is the world (static). Dynamic links are 1..nl, parents before children.
 *     Bodies joined by `fixed` joints are merged into their first non-fixed ancestor's link.
 *   - link frame == joint frame after the joint motion:  E_0i = E_0parent * E_pj0 * Q(q_joint)
 *   - poses are (R row-major 3x3, p): x_parent = R x_child + p ; quaternions in the XML are w x y z
 *   - spatial vectors are (angular; linear), world-frame, referred to the world origin.
 */
#ifndef TSIM_BLOB_H
#define TSIM_BLOB_H

#define TSIM_MAGIC   0x7531494D
#define TSIM_VERSION 3

/* ---- int header (I[0..TSIM_IH_SIZE)) ---- */
enum {
  TSIM_IH_MAGIC = 0, TSIM_IH_VERSION, TSIM_IH_NL, TSIM_IH_NR, TSIM_IH_NU, TSIM_IH_NVAR,
  TSIM_IH_NPAIR, TSIM_IH_NCPT, TSIM_IH_NSENSOR, TSIM_IH_NTAXEL, TSIM_IH_NSPRIM,
  TSIM_IH_INTEGRATOR, TSIM_IH_MAX_ITER, TSIM_IH_MAX_LS,
  /* offsets into I[] */
  TSIM_IH_OFF_LINK, TSIM_IH_OFF_DOF, TSIM_IH_OFF_MOTOR, TSIM_IH_OFF_VAR, TSIM_IH_OFF_PAIR,
  TSIM_IH_OFF_SENSOR, TSIM_IH_OFF_SPRIM,
  /* offsets into F[] */
  TSIM_IH_FOFF_LINK, TSIM_IH_FOFF_DOF, TSIM_IH_FOFF_MOTOR, TSIM_IH_FOFF_VAR, TSIM_IH_FOFF_PAIR,
  TSIM_IH_FOFF_SENSOR, TSIM_IH_FOFF_CPT, TSIM_IH_FOFF_TAXEL,
  TSIM_IH_NI, TSIM_IH_NF,           /* total lengths of I[] and F[] */
  TSIM_IH_NDOF_TACTILE,             /* 3 * ntaxel */
  TSIM_IH_SIZE = 40
};

/* The Newton iteration of a sub-step is fully specified by the model's <solver_option>: TSIM_FH_TOL, TSIM_IH_MAX_ITER, TSIM_IH_MAX_LS
 * (backtracking halvings per iteration).  No other solver constant exists: kernels and oracle run that loop literally. */

/* ---- float header (F[0..TSIM_FH_SIZE)) ---- */
enum { TSIM_FH_H = 0, TSIM_FH_GX, TSIM_FH_GY, TSIM_FH_GZ, TSIM_FH_TOL, TSIM_FH_SIZE = 8 };

/* joint types */
enum { TSIM_J_REVOLUTE = 1, TSIM_J_PRISMATIC = 2, TSIM_J_PLANAR = 3, TSIM_J_TRANSLATIONAL = 4,
       TSIM_J_FREE3D_EULER = 5, TSIM_J_FREE3D_EXP = 6,   /* 5, 6: never emitted — the compiler decomposes them */
       TSIM_J_SPHERICAL_EXP = 7 };                         /* 3 dofs: rotation vector theta, R = exp([theta]) */

/* link record: ints */
enum { TSIM_LI_PARENT = 0, TSIM_LI_JTYPE, TSIM_LI_DOF0, TSIM_LI_NDOF, TSIM_LI_ANCMASK, TSIM_LI_SIZE = 8 };
/* link record: floats — E_pj0 (R 9, p 3), axes (3x3, row k = axis k in joint frame),
 * mass, com (link frame), inertia about com in link frame (xx yy zz xy xz yz) */
enum { TSIM_LF_R = 0, TSIM_LF_P = 9, TSIM_LF_AXES = 12, TSIM_LF_MASS = 21, TSIM_LF_COM = 22,
       TSIM_LF_INERTIA = 25, TSIM_LF_SIZE = 32 };

/* dof record */
enum { TSIM_DI_LINK = 0, TSIM_DI_SIZE = 2 };
enum { TSIM_DF_DAMPING = 0, TSIM_DF_LIM_LO, TSIM_DF_LIM_HI, TSIM_DF_LIM_K, TSIM_DF_SIZE = 4 };

/* motor record (one per entry of u) */
enum { TSIM_MI_DOF = 0, TSIM_MI_CTRL, TSIM_MI_SIZE = 2 };      /* ctrl: 0 force, 1 position */
enum { TSIM_MF_LO = 0, TSIM_MF_HI, TSIM_MF_P, TSIM_MF_D, TSIM_MF_SIZE = 4 };

/* variable (end-effector point) record */
enum { TSIM_VI_LINK = 0, TSIM_VI_SIZE = 1 };
enum { TSIM_VF_POS = 0, TSIM_VF_SIZE = 3 };

/* primitive shapes */
enum { TSIM_P_PLANE = 0, TSIM_P_CUBOID = 1, TSIM_P_SPHERE = 2, TSIM_P_CYLINDER = 3 };

/* contact pair record: points fixed on link A tested against a primitive fixed on link B */
enum { TSIM_PI_LINKA = 0, TSIM_PI_LINKB, TSIM_PI_PRIM, TSIM_PI_PT0, TSIM_PI_NPT, TSIM_PI_FLAGS,
       TSIM_PI_SIZE = 6 };
/* flags bit0: dynamics-active (0 = sensing-only pair, used only by tactile sensors)
 *       bit1: moving contact point = lowest point of sphere A on plane B (sphere-ground)   */
/* floats: primitive frame in link B (R 9, p 3), shape params (4), kn kt mu kd */
enum { TSIM_PF_R = 0, TSIM_PF_P = 9, TSIM_PF_SHAPE = 12, TSIM_PF_KN = 16, TSIM_PF_KT, TSIM_PF_MU,
       TSIM_PF_KD, TSIM_PF_SIZE = 20 };
/* shape params: cuboid = half sizes xyz ; sphere = radius ; cylinder = radius, half length (axis z);
 *               plane = none (normal is the frame's +z)                                    */

/* tactile sensor record */
enum { TSIM_SI_LINK = 0, TSIM_SI_TAX0, TSIM_SI_NTAX, TSIM_SI_SPRIM0, TSIM_SI_NSPRIM, TSIM_SI_ROWS,
       TSIM_SI_COLS, TSIM_SI_SIZE = 8 };
enum { TSIM_SF_KN = 0, TSIM_SF_KT, TSIM_SF_MU, TSIM_SF_KD, TSIM_SF_SIZE = 4 };
/* sensor-primitive list: I[off_sprim + j] = index of the contact pair whose primitive is tested */

/* contact points: F[foff_cpt + c*ncpt + i], c = 0..2  (SoA: x[], y[], z[] in link-A frame)        */
/* taxels:         F[foff_taxel + c*ntaxel + i], c = 0..11 (SoA: pos xyz, axis0 xyz, axis1 xyz,
 *                 normal xyz, all in the sensor link's frame)                                    */

#endif
