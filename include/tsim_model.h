/* tsim_model.h — the model loader behind the C ABI (host code, no GPU work).
 *
 * The reference constructs a simulation from a model file:  redmax_py.Simulation(model_path)
 * (envs/redmax_torch_env.py:33, examples/RollingBallExp/test_sim_speed.py:35) and edits it through the update_* family
 * (SURVEY.md §8b).  A C / C++ host gets the same two things here: tsim_model_load reads the redmax XML (and the OBJ meshes, contact-point
 * and taxel files it names, relative to the XML's directory) and compiles it to the flat blob of include/tsim_blob.h that
 * tsim_batch_create takes; tsim_model_update edits the description and recompiles (-> tsim_update_model on the batches that use it).
 * It is the native counterpart of tactilesimulation_amd/model/compiler.py: the two produce the same ints and the same reals to
 * round-off of the host's double arithmetic (tests/test_native_model_loader.py).
 *
 * Every function returns 0 on success unless stated otherwise; on failure tsim_last_error() (include/tsim.h) says why.  A tsim_model is
 * used from one thread at a time. */
#ifndef TSIM_MODEL_H
#define TSIM_MODEL_H

#include <stdint.h>
#include "tsim.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tsim_model tsim_model;

/* redmax_py.Simulation(model_path)      envs/redmax_torch_env.py:33 */
int tsim_model_load(const char* xml_path, tsim_model** out);
void tsim_model_free(tsim_model* m);

/* The compiled blob: I[nI] / F[nF] as tsim_batch_create takes them.  The pointers stay valid until the next tsim_model_update /
 * tsim_model_free of this model. */
int tsim_model_blob(const tsim_model* m, const int32_t** I, int* nI, const double** F, int* nF);

/* A compiled blob as a file, for hosts that ship models without their XML / mesh files: little-endian
 * { uint32 TSIM_MAGIC, uint32 TSIM_VERSION, int32 nI, int32 nF, int32 I[nI], float64 F[nF] }.
 * A model loaded from a blob file has no description: tsim_model_update and tsim_model_image_pos fail on it. */
int tsim_model_save_blob(const tsim_model* m, const char* path);
int tsim_model_load_blob(const char* path, tsim_model** out);

/* tsim_batch_create on this model's blob */
int tsim_batch_create_from_model(const tsim_model* m, int B, int tape_capacity, int dtype, int device, tsim_batch** out);

/* sim.get_tactile_image_pos(name) -> [(row, col)] per taxel of one sensor      examples/RollingBallExp/test_sim_speed.py:57-61,
 * envs/dclaw_rotate_env.py:69-72.  Writes min(count, capacity) (row, col) pairs to rc_out[2 * capacity] and RETURNS the sensor's taxel
 * count (-1: unknown sensor / no description). */
int tsim_model_image_pos(const tsim_model* m, const char* sensor_name, int32_t* rc_out, int capacity);

/* The update_* family (envs/dclaw_rotate_env.py:173-178, envs/stable_grasp_env.py:122,
 * envs/tactile_insertion_env.py:254-279, envs/tactile_push_env.py:148-152): edit the description, recompile.
 *   what                              name                 name2            values[n]
 *   TSIM_UPD_JOINT_DAMPING            joint                -                damping
 *   TSIM_UPD_JOINT_LOCATION           joint                -                x y z
 *   TSIM_UPD_BODY_DENSITY             body                 -                density
 *   TSIM_UPD_BODY_SIZE                body                 -                cuboid: sx sy sz | sphere: radius | cylinder: length radius
 *   TSIM_UPD_ENDEFFECTOR_POSITION     end-effector         -                x y z
 *   TSIM_UPD_CONTACT_PARAMETERS       general body         primitive body   kn kt mu damping  (NaN: keep)
 *   TSIM_UPD_TACTILE_PARAMETERS       sensor body or name  -                kn kt mu damping  (NaN: keep)
 *   TSIM_UPD_VIRTUAL_OBJECT           virtual object       -                x y z qw qx qy qz (viewer only: stored, nothing simulated changes)
 * On failure the model is unchanged. */
enum { TSIM_UPD_JOINT_DAMPING = 0, TSIM_UPD_JOINT_LOCATION, TSIM_UPD_BODY_DENSITY, TSIM_UPD_BODY_SIZE, TSIM_UPD_ENDEFFECTOR_POSITION,
       TSIM_UPD_CONTACT_PARAMETERS, TSIM_UPD_TACTILE_PARAMETERS, TSIM_UPD_VIRTUAL_OBJECT };
int tsim_model_update(tsim_model* m, int what, const char* name, const char* name2, const double* values, int n);

/* Index into F[] (= column of a tsim_set_env_tables row) of one numeric parameter, for per-environment randomisation; RETURNS the index,
 * -1 if there is no such record.
 *   TSIM_TAB_PAIR    key0 = general body or "ground", key1 = primitive body (ground pair: the body);  field 0 kn 1 kt 2 mu 3 damping, 4..7 shape 0..3
 *   TSIM_TAB_SENSOR  key0 = sensor name;                                                               field 0 kn 1 kt 2 mu 3 damping
 *   TSIM_TAB_DOF     key0 = joint name, field = dof of the joint (0 ..);                               the dof's damping */
enum { TSIM_TAB_PAIR = 0, TSIM_TAB_SENSOR = 1, TSIM_TAB_DOF = 2 };
int tsim_model_table_offset(const tsim_model* m, int kind, const char* key0, const char* key1, int field);

#ifdef __cplusplus
}
#endif
#endif
