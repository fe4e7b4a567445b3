// Internal structures of the U-Net engine shared by unet_api.hip (planner, forward) and train.hip (backward,
// optimizer).  Not part of the C ABI.
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "conv_igemm.h"
#include "conv_rh.h"
#include "conv_h16.h"
#include "conv_box.h"
#include "first_pass.h"
#include "wino.h"

namespace bsmi {

struct HostWeight {
  std::vector<int64_t> shape;
  std::vector<float> data;
  bool loaded = false;
};

// One 32-byte unit of K: 16 (bf16) / 8 (f32) consecutive channels of one kernel tap of one
// source tensor.  kUnitsPerStep units of the same tensor slot form a K-step.
struct PackEntry {
  int slot;        // tensor slot of the launch
  int dz, dy, dx;  // tap offset (voxels) relative to the slot's origin
  int c0;          // first channel of the unit
  int wsrc;        // 0 = this stage's conv weight, 1 = residual 1x1x1 weight
  int tap;         // flat tap index into the weight's kernel dims
  int cin_base;    // first input channel of this slot inside the weight's Cin dim
  int creal;       // real channels of the slot
  bool dummy;      // filler: delta 0, all-zero weights
  int phase;       // index into PackedConv::phases
};

// A phase groups the units of one (slot, 32-channel chunk): kind 0 = all kernel taps, kind 1 = the
// residual tap (bookkeeping of build_entries; conv_rh.hip splits kind 0 further by z-tap).
struct PackPhase {
  int slot, c0, kind, first_unit, nunits;
};

struct PackedConv {
  bool ready = false;
  std::vector<PackEntry> entries;
  std::vector<PackPhase> phases;
  void* w = nullptr;
  int ks = 1;                 // kernel K-steps per logical K-step (3: the split mode as listed K-steps)
  size_t lo_image_bytes = 0;  // fused split-bf16: the lo image follows the hi image at this byte offset (0: no lo image)
  float* bias = nullptr;
  int Npad = 0;
  TileCfg tile = TILE_256x32;
};

// Winograd form of one 3x3x3 stage in the split-bf16 mode (wino.hip): transformed weights of the 16 batched GEMMs and, for the
// last stage of a ConvPass, the weights of the cropped 1x1x1 residual branch as a launch of its own (raw sums, added by the
// output transform).
struct PackedWino {
  bool ready = false;
  int m = 2;                        // output tile edge: F(2x2, 3x3) or F(4x4, 3x3)
  std::vector<WinoUnit> units;      // K-step list of every batch
  std::vector<int> cin_of_v;        // V channel -> input channel of the concatenated input (-1: pad)
  int Cv = 0;                       // channels of V: every source padded to kChanPad
  void* w = nullptr;                // hi image, lo image
  size_t lo_image_bytes = 0, batch_bytes = 0;
  int Npad = 0;
  TileCfg tile = TILE_256x32;
  std::vector<PackEntry> res_entries;  // units of the residual-only launch (empty: no residual branch on this stage)
  void* res_w = nullptr;
  size_t res_lo_image_bytes = 0;
  // the same branch cut in two for a ConvPass whose second source is an upsampled map (unet.py:215-223): [0] the units of the
  // skip connection (a launch at full resolution), [1] those of the upsampled map, which then run BELOW the upsampling
  // (plan_wino, fuse_up); each list padded to an even number of K-steps, each with its own weight images
  std::vector<PackEntry> res_part[2];
  void* res_part_w[2] = {nullptr, nullptr};
  size_t res_part_lo[2] = {0, 0};
};

// Halo-resident form of a stage with at most 64 output channels in the split-bf16 mode (conv_h16.hip): its own K-step list --
// per (source, 16-channel chunk, z tap) a phase whose K-steps pair two in-plane taps -- and weight images.
struct H16PhaseHost {
  int slot, c0, dz, kind;  // kind 1: the residual's single tap (dz, dy, dx of its entry = the crop centre)
  int first_step, nsteps;
};
struct PackedH16 {
  bool ready = false;
  std::vector<PackEntry> entries;  // two units per K-step: (tap a, c0), (tap b, c0) or a dummy
  std::vector<H16PhaseHost> phases;
  void* w = nullptr;               // hi image, lo image
  size_t lo_image_bytes = 0;
  int Npad = 0;                    // 16 or 64
};

struct PassSite {
  std::string prefix;
  int nslots = 1;
  int cin[2] = {0, 0};
  int cout = 0;
  int nconv = 0;
  int k[BSMI_MAX_CONVS][3];
  PackedConv packed[BSMI_NUM_PREC][BSMI_MAX_CONVS];
  PackedWino wino[BSMI_MAX_CONVS];  // BSMI_PREC_BF16X3 only
  PackedH16 h16[BSMI_MAX_CONVS];    // BSMI_PREC_BF16X3 only
};

struct HeadSite {
  std::string prefix;
  int cin = 0, cout = 0;
  float* hw = nullptr;  // device [cout][2][cin]
  float* hb = nullptr;  // device [cout][2]
};

struct TDesc {
  void* ptr = nullptr;
  int C = 0, Cpad = 0, D = 0, H = 0, W = 0;
  size_t lo_off = 0;  // BSMI_PREC_BF16X3: 16 = each 16-byte vector of hi values is followed by that of the lo values; else 0
};

struct PlanStep {
  enum Type { INPUT, CONV, POOL, UP, HEAD } type;
  ConvArgs conv;
  RhArgs rh;
  bool use_rh = false;
  H16Args h16;           // halo-resident split-bf16 launch of a narrow stage (use_h16)
  bool use_h16 = false;
  int h16_rows = 0;      // rows of its halo buffer
  BoxArgs box;           // conv_box.hip launch of this step (use_box)
  bool use_box = false;
  // Winograd form of this step (use_wino): input transform, batched GEMMs, [residual-only launch], output transform
  bool use_wino = false;
  WinoInArgs wino_in;
  ConvArgs wino_gemm, wino_res, wino_res_low;
  bool wino_has_res = false, wino_has_res_low = false;
  WinoOutArgs wino_out;
  TileCfg tile;
  TDesc in, out;
  int f[3], o[3];
  bool skip = false;     // UP step whose map nobody reads: its consumers upsample on the fly (plan_wino, fuse_up)
  int head = 0;
  double flops = 0;  // algorithmic FLOPs of this launch
  // multiply-adds the matrix pipe is actually given (x 2): tile-padded rows x padded columns x padded K, every batch of a Winograd
  // stage, three bf16 products per one of the split mode -- what `roofline.executed_mfma_frac` of bench.py prices against the
  // dense bf16 peak.  Exact for the implicit-GEMM and Winograd launches; the halo / box / first-pass kernels count their
  // algorithmic products (x 3 in the split mode), i.e. without their padding
  double exec_flops = 0;
  // what the backward pass (train.hip) needs to know about a CONV step
  struct PassSite* site = nullptr;
  int ci = 0;                        // stage index inside the ConvPass
  TDesc slots[kMaxConvTensors];      // source tensors of the launch
  int so[kMaxConvTensors][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};  // their origins (voxels)
  int nsl = 0;
  // training (train.hip): this launch as a fused split-bf16 one over split copies of its f32 sources; used only while
  // bsmi_unet::train_forward is set, so the f32 inference mode of the same plan stays exact f32
  struct TrainFwdX3* tx3 = nullptr;
};

struct Plan {
  std::vector<float> last_ms;  // per-step durations of the last harvested forward
  bool pending = false;        // events recorded but not yet harvested
  std::vector<void*> allocs;
  std::vector<PlanStep> steps;
  int64_t out_shape[3] = {0, 0, 0};
  double flops = 0;
  size_t bytes = 0;
  // Profiling: a set of 2 events per step for every profiled forward that has not been read yet (`inflight`, oldest
  // first), so that recording never waits for an earlier forward; read sets go back to `spare`.  `events` is the set of
  // the forward being recorded.
  std::vector<hipEvent_t> events;
  std::vector<std::vector<hipEvent_t>> inflight, spare;
  bool profiled = false;           // last forward recorded events
  bool fused_first = false;        // steps 0..2 (INPUT, CONV, CONV of l_conv.0) run as one first_pass launch
  int prec = 0;                    // precision the plan was built for
};

struct TrainState;

int esize(int prec);  // bytes of one stored element (per plane in the split mode)
int ksplit(int prec); // K-steps the kernel walks per logical K-step (3 in the split-bf16 mode)
int bke(int prec);   // elements per K-step row
int sube(int prec);  // elements per 32-byte unit
// unit list of stage `ci` of a ConvPass (see unet_api.hip)
void build_entries(const PassSite& p, int ci, int prec, std::vector<PackEntry>& out, std::vector<PackPhase>* phases_out = nullptr);

}  // namespace bsmi

using namespace bsmi;

struct bsmi_unet {
  bsmi_unet_config cfg;
  int device = 0;
  int nl = 0;
  std::vector<PassSite> l_conv, r_conv;
  std::vector<HeadSite> heads;
  int crop_factor[BSMI_MAX_LEVELS][3];
  std::map<std::string, HostWeight> weights;
  bool finalized[BSMI_NUM_PREC] = {false, false, false};
  int profile_period = 0;       // 0 off, N: every Nth forward is timed step by step
  uint64_t profile_count = 0;
  Plan* last_plan = nullptr;
  double prof_ms[5] = {0, 0, 0, 0, 0}, prof_flops[5] = {0, 0, 0, 0, 0};
  double prof_exec = 0;  // executed FLOPs of the profiled conv launches (PlanStep::exec_flops)
  int64_t prof_launches[5] = {0, 0, 0, 0, 0};
  std::map<std::vector<int64_t>, std::unique_ptr<Plan>> plans;  // key: prec, D, H, W
  float* sk_ws = nullptr;  // split-K tail partial tiles + work-queue counters (conv_igemm.h)
  int sk_grid = 0;         // 0: not set up yet, -1: disabled
  int sk_request = -1;     // bsmi_unet_set_persistent_grid: -1 = CU count of the device, 0 = off
  bsmi::TrainState* train = nullptr;  // train.hip
  bool train_forward = false;         // the forward pass of a training step is running (PlanStep::tx3)
  int train_split = 1;                // bsmi_unet_train_set_arithmetic
  int train_det = 0;                  // bsmi_unet_train_set_deterministic: ordered reductions instead of float atomics
  bsmi::FirstPassWeights first_pass;     // first_pass.hip: weights of the fused first ConvPass (bf16 mode)
  bsmi::FirstPassWeights first_pass_x3;  // ... and of the split-bf16 mode
};

namespace bsmi {
// the cached launch plan of (precision, input shape); built on first use
int get_plan(bsmi_unet* h, int precision, const int64_t in_shape[3], Plan** out);
void free_train_state(bsmi_unet* h);
int train_forward_conv_x3(bsmi_unet* h, const PlanStep& st, hipStream_t s);  // train.hip
int train_refresh_f32_images(bsmi_unet* h, hipStream_t s);  // train.hip: f32 weight images left stale by the last optimizer step
}  // namespace bsmi
