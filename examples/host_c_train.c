/* host_c_train.c -- a host with NO tensor library driving libvisdial_hip.so through the model-level C ABI only.
 *
 * This is the call sequence of lua/model.lua (Model:__init, Model:upload, Model:trainIteration with its software-pipelined
 * prefetch, Model:initComm, Model:retrieveBatch), written in C because no Lua/LuaJIT exists in the build container or on the GPU
 * box: what LuaJIT's ffi does -- dlopen the library, declare include/visdial_hip.h, pass host pointers -- is exactly what this
 * file does, so it is the executable stand-in for the Lua host (tests/test_abi_c_host_gpu.py builds it with gcc and compares its
 * losses and ranks with the Python host driving the same ABI on the same batches).
 *
 *   gcc -O2 -I include examples/host_c_train.c -ldl -o host_c_train
 *   ./host_c_train <libvisdial_hip.so> <batches.bin> <steps> <use_comm 0|1>
 *
 * batches.bin (little endian, written by the test): header int32 {B, R, O, Tq, Th, To, S2xC floats per image, vocabSize, embedSize,
 * rnnHiddenSize, imgFeatureSize, imgSpatialSize, commonEmbeddingSize, nbatches}, then per batch: ques_fwd [B*R*Tq] int32,
 * hist [B*R*Th] int32, img_feat [B*S2xC] float, options [B*R*O*To] int32, answer_ind [B*R] int32.
 * Output (stdout): "rank <n> <gt rank>" for the first batch under the initial parameters, then one line "loss <step> <value>" per
 * training step (vd_model_loss) and the learning rate. */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "visdial_hip.h"

#define SYM(name) name##_t p_##name = (name##_t)dlsym(lib, #name); if (!p_##name) { fprintf(stderr, "missing symbol %s\n", #name); return 2; }
#define CHECK(call) do { int rc__ = (call); if (rc__ != 0) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc__, p_vd_last_error()); return 3; } } while (0)

typedef const char* (*vd_last_error_t)(void);
typedef int (*vd_set_device_t)(int);
typedef int (*vd_model_create_t)(const vd_model_params*, const char*, const char*, vd_model**);
typedef void (*vd_model_destroy_t)(vd_model*);
typedef int (*vd_model_init_params_t)(vd_model*, uint64_t);
typedef int (*vd_model_set_training_t)(vd_model*, int);
typedef int (*vd_model_upload_batch_t)(vd_model*, const vd_batch*);
typedef int (*vd_model_forward_backward_t)(vd_model*, int);
typedef int (*vd_model_loss_t)(vd_model*, float*);
typedef int (*vd_model_update_t)(vd_model*, float);
typedef int (*vd_model_learning_rate_t)(vd_model*, double*, int);
typedef int (*vd_model_retrieve_t)(vd_model*);
typedef int (*vd_model_ranks_t)(vd_model*, int, int32_t*);
typedef int (*vd_comm_unique_id_t)(void*);
typedef int (*vd_comm_init_t)(int, int, const void*);
typedef int (*vd_comm_destroy_t)(void);
typedef int (*vd_model_allreduce_grads_t)(vd_model*);
typedef int (*vd_model_synchronize_t)(vd_model*);

static void* read_exact(FILE* f, size_t bytes) {
  void* p = malloc(bytes ? bytes : 1);
  if (!p || fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "short read (%zu bytes)\n", bytes); exit(4); }
  return p;
}

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: %s <lib.so> <batches.bin> <steps> <use_comm>\n", argv[0]); return 1; }
  void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
  SYM(vd_last_error) SYM(vd_set_device) SYM(vd_model_create) SYM(vd_model_destroy) SYM(vd_model_init_params)
  SYM(vd_model_set_training) SYM(vd_model_upload_batch) SYM(vd_model_forward_backward) SYM(vd_model_loss) SYM(vd_model_update)
  SYM(vd_model_learning_rate) SYM(vd_model_retrieve) SYM(vd_model_ranks) SYM(vd_comm_unique_id) SYM(vd_comm_init)
  SYM(vd_comm_destroy) SYM(vd_model_allreduce_grads) SYM(vd_model_synchronize)
  const int steps = atoi(argv[3]), use_comm = atoi(argv[4]);
  FILE* f = fopen(argv[2], "rb");
  if (!f) { perror(argv[2]); return 4; }
  int32_t h[14];
  if (fread(h, sizeof(int32_t), 14, f) != 14) return 4;
  const int B = h[0], R = h[1], O = h[2], Tq = h[3], Th = h[4], To = h[5], img_floats = h[6], nb = h[13];
  const int N = B * R;

  /* Model:__init (lua/model.lua): params -> vd_model_create(encoder name, decoder name) -> init */
  vd_model_params p;
  memset(&p, 0, sizeof(p));
  p.vocabSize = h[7]; p.embedSize = h[8]; p.rnnHiddenSize = h[9]; p.imgFeatureSize = h[10]; p.imgSpatialSize = h[11];
  p.commonEmbeddingSize = h[12]; p.numAttentionLayers = 1; p.maxQuesCount = R; p.numOptions = O;
  p.learningRate = 1e-3f; p.lrDecayRate = 0.9997592083f; p.minLRate = 5e-5f; p.seed = 1234; p.lstmBf16 = 0; p.useStreams = 1;
  p.numLayers = 2; p.imgEmbedSize = 300; p.dropout = 0.5f;
  CHECK(p_vd_set_device(0));
  vd_model* m = NULL;
  CHECK(p_vd_model_create(&p, "mn-att-ques-im-hist", "disc", &m));
  CHECK(p_vd_model_init_params(m, 1234));
  CHECK(p_vd_model_set_training(m, 0));          /* dropout off: the comparison run does the same */
  if (use_comm) {                                 /* Model.commUniqueId / Model:initComm with world 1 */
    char id[128];
    CHECK(p_vd_comm_unique_id(id));
    CHECK(p_vd_comm_init(0, 1, id));
  }
  /* the batches (a dataloader would hand them out one by one) */
  vd_batch* bt = (vd_batch*)calloc((size_t)nb, sizeof(vd_batch));
  for (int i = 0; i < nb; ++i) {
    bt[i].B = B; bt[i].Tq = Tq; bt[i].Th = Th; bt[i].To = To;
    bt[i].ques_fwd = (const int32_t*)read_exact(f, (size_t)N * Tq * 4);
    bt[i].hist = (const int32_t*)read_exact(f, (size_t)N * Th * 4);
    bt[i].img_feat = (const float*)read_exact(f, (size_t)B * img_floats * 4);
    bt[i].options = (const int32_t*)read_exact(f, (size_t)N * O * To * 4);
    bt[i].answer_ind = (const int32_t*)read_exact(f, (size_t)N * 4);
  }
  fclose(f);
  /* Model:retrieveBatch on the first batch (initial parameters): ground-truth ranks */
  CHECK(p_vd_model_upload_batch(m, &bt[0]));
  CHECK(p_vd_model_retrieve(m));
  int32_t* ranks = (int32_t*)malloc((size_t)N * sizeof(int32_t));
  CHECK(p_vd_model_ranks(m, 1, ranks));
  for (int n = 0; n < N; ++n) printf("rank %d %d\n", n, ranks[n]);
  /* Model:trainIteration: enqueue the step, upload the NEXT batch while the device runs, read this step's loss */
  CHECK(p_vd_model_upload_batch(m, &bt[0]));
  for (int it = 0; it < steps; ++it) {
    CHECK(p_vd_model_forward_backward(m, 0));
    if (use_comm) CHECK(p_vd_model_allreduce_grads(m));    /* enqueue only; world 1: the RCCL kernels still run */
    CHECK(p_vd_model_update(m, 1.0f));
    CHECK(p_vd_model_upload_batch(m, &bt[(it + 1) % nb]));
    float loss = 0.f;
    CHECK(p_vd_model_loss(m, &loss));
    printf("loss %d %.9g\n", it, loss);
  }
  double lr = 0;
  CHECK(p_vd_model_learning_rate(m, &lr, 0));
  printf("lr %.12g\n", lr);
  CHECK(p_vd_model_synchronize(m));
  if (use_comm) CHECK(p_vd_comm_destroy());
  p_vd_model_destroy(m);
  return 0;
}
