/* A plain C host of the HIP path: what a C / C++ maintainer writes where the reference's Python writes
 *     sim = redmax_py.Simulation(model_path); sim.reset(True); sim.set_u(u); sim.forward(5); q = sim.get_q(); ...; sim.backward_steps(5)
 * (envs/redmax_torch_env.py:33, envs/redmax_torch_functions.py:115-170) — for B environments at once, with nothing but the C ABI of
 * include/tsim.h + include/tsim_model.h and the HIP runtime for device memory.  No Python, no torch.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_host/step_from_xml.c -o step_from_xml \
 *       -Ltactilesimulation_amd/csrc -ltsim_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/tactilesimulation_amd/csrc -Wl,-rpath,/opt/rocm/lib
 *   ./step_from_xml tests/models/slider_push.xml 8 6 0.6            (a 5th argument "general_body:primitive_body": one contact stiffness per environment)
 *
 * Prints, per env-step, q of environment 0 and B-1 (fp64, %.17g), then dL/du of L = sum(q_final) for environment 0. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "tsim.h"
#include "tsim_model.h"

#define OK(call) do { if ((call) != 0) { fprintf(stderr, "%s: %s\n", #call, tsim_last_error()); return 1; } } while (0)
#define HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: %s model.xml B steps u_value [general_body:primitive_body]\n", argv[0]); return 2; }
  const int B = atoi(argv[2]), T = atoi(argv[3]), S = 5;
  const double uval = atof(argv[4]);

  tsim_model* model = NULL;
  OK(tsim_model_load(argv[1], &model));
  tsim_batch* sim = NULL;
  OK(tsim_batch_create_from_model(model, B, T * S, TSIM_F64, 0, &sim));
  const int nr = tsim_ndof_r(sim), nu = tsim_ndof_u(sim), nvar = tsim_ndof_var(sim), ntac = tsim_ndof_tactile(sim);
  printf("model %s: ndof_r %d ndof_u %d ndof_var %d ndof_tactile %d h %.17g kernels %s\n", argv[1], nr, nu, nvar, ntac, tsim_timestep(sim), tsim_kernel_variant(sim));

  double *q0, *u, *q, *var, *tac, *seed, *dldu;
  int32_t* status;
  HIP(hipMalloc((void**)&q0, sizeof(double) * B * nr));   HIP(hipMalloc((void**)&u, sizeof(double) * B * (nu ? nu : 1)));
  HIP(hipMalloc((void**)&q, sizeof(double) * B * nr));    HIP(hipMalloc((void**)&var, sizeof(double) * B * (nvar ? nvar : 1)));
  HIP(hipMalloc((void**)&tac, sizeof(double) * B * (ntac ? ntac : 1))); HIP(hipMalloc((void**)&status, sizeof(int32_t) * B));
  HIP(hipMalloc((void**)&seed, sizeof(double) * B * nr)); HIP(hipMalloc((void**)&dldu, sizeof(double) * B * T * S * (nu ? nu : 1)));

  double* host = (double*)calloc((size_t)B * (nr > nu ? nr : nu) * T * S + 1, sizeof(double));
  HIP(hipMemcpy(q0, host, sizeof(double) * B * nr, hipMemcpyHostToDevice));                 /* q0 = 0 */
  for (int e = 0; e < B; ++e) for (int k = 0; k < nu; ++k) host[e * nu + k] = uval * (1.0 + 0.1 * e);      /* one control per environment */
  HIP(hipMemcpy(u, host, sizeof(double) * B * (nu ? nu : 1), hipMemcpyHostToDevice));

  double* tables = NULL;
  if (argc > 5) {
    /* Domain randomisation, one parameter set per environment (what envs/tactile_insertion_env.py:238-281 draws per episode through
     * update_contact_parameters, here for the whole batch at once): rows = copies of the model's numeric tables, one entry overwritten. */
    char key[256]; strncpy(key, argv[5], sizeof(key) - 1); key[sizeof(key) - 1] = 0;
    char* colon = strchr(key, ':');
    if (!colon) { fprintf(stderr, "expected general_body:primitive_body\n"); return 2; }
    *colon = 0;
    const int col = tsim_model_table_offset(model, TSIM_TAB_PAIR, key, colon + 1, 0);          /* field 0: kn */
    if (col < 0) { fprintf(stderr, "%s\n", tsim_last_error()); return 1; }
    const int32_t* I; const double* F; int nI, nF;
    OK(tsim_model_blob(model, &I, &nI, &F, &nF));
    const int n = tsim_table_size(sim);
    double* rows = (double*)malloc(sizeof(double) * (size_t)B * n);
    for (int e = 0; e < B; ++e) { memcpy(rows + (size_t)e * n, F, sizeof(double) * n); rows[(size_t)e * n + col] = F[col] * (1.0 + 0.05 * e); }
    HIP(hipMalloc((void**)&tables, sizeof(double) * (size_t)B * n));
    HIP(hipMemcpy(tables, rows, sizeof(double) * (size_t)B * n, hipMemcpyHostToDevice));
    OK(tsim_set_env_tables(sim, tables, NULL));
    printf("per-environment tables: column %d (kn of %s -> %s) = %.17g x (1 + 0.05 e)\n", col, key, colon + 1, F[col]);
    free(rows);
  }
  OK(tsim_reset(sim, q0, NULL, 1, NULL));                                                    /* set_state_init + reset(backward_flag=True) */
  for (int t = 0; t < T; ++t) {
    OK(tsim_step(sim, u, S, q, NULL, nvar ? var : NULL, ntac ? tac : NULL, status, NULL));   /* set_u; forward(5); get_q; get_variables; get_tactile_force_vector */
    HIP(hipDeviceSynchronize());
    HIP(hipMemcpy(host, q, sizeof(double) * B * nr, hipMemcpyDeviceToHost));
    printf("step %d q[0]", t);
    for (int k = 0; k < nr; ++k) printf(" %.17g", host[k]);
    printf(" | q[%d]", B - 1);
    for (int k = 0; k < nr; ++k) printf(" %.17g", host[(B - 1) * nr + k]);
    printf("\n");
  }
  int32_t* st = (int32_t*)malloc(sizeof(int32_t) * B);
  HIP(hipMemcpy(st, status, sizeof(int32_t) * B, hipMemcpyDeviceToHost));
  int bad = 0; for (int e = 0; e < B; ++e) bad += st[e] != 0;
  printf("non-converged environments in the last step: %d\n", bad);

  for (int i = 0; i < B * nr; ++i) host[i] = 1.0;                                            /* dL/dq_final = 1 */
  HIP(hipMemcpy(seed, host, sizeof(double) * B * nr, hipMemcpyHostToDevice));
  OK(tsim_backward_steps(sim, T * S, 0, seed, NULL, NULL, dldu, NULL));                      /* backward_info.df_dq = ...; backward_steps(n); backward_results.df_du */
  HIP(hipDeviceSynchronize());
  HIP(hipMemcpy(host, dldu, sizeof(double) * T * S * (nu ? nu : 1), hipMemcpyDeviceToHost));
  printf("dL/du[0]");
  for (int i = 0; i < T * S * nu; ++i) printf(" %.17g", host[i]);
  printf("\n");

  tsim_batch_destroy(sim);
  tsim_model_free(model);
  free(host); free(st);
  if (tables) hipFree(tables);
  hipFree(q0); hipFree(u); hipFree(q); hipFree(var); hipFree(tac); hipFree(status); hipFree(seed); hipFree(dldu);
  return 0;
}
