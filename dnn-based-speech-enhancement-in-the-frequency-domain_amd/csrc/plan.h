// Host-side plan: buffers, constants and op lists for one model configuration (pure C++, no HIP).
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>
#include "sefd_desc.h"

namespace sefd {

struct ModelConfig {
  int32_t model;            // 0 = DCCRN, 1 = CRN
  int32_t B, L;
  int32_t win_len, hop, fft_len;
  int32_t n_layers;
  int32_t kernel_num[12];   // output channels per encoder layer (complex: real+imag); FullSubNet: see build_fsn_plan
  int32_t rnn_layers, rnn_units;
  int32_t mask_mode;        // 0 E, 1 C, 2 R
  int32_t lstm_complex;     // 1 = NavieComplexLSTM stack, 0 = real nn.LSTM(2 layers)+Linear
  int32_t skip;             // skip connections (cfg.skip_type)
  int32_t act_dtype;        // DT_F32 / DT_BF16 storage + MFMA dtype of the conv stack
  int32_t kernel_size;      // 5
  int32_t training;         // 1: BatchNorm batch statistics + running update ; 0: eval (running stats)
  int32_t bn_world;         // > 1: SyncBN over that many ranks (sync points, counts scaled)
  int32_t grad_buckets;     // 2: decoder + LSTM gradients unpacked before the encoder backward (DDP overlap)
  int32_t use_cbn;          // DCCRN: ComplexBatchNorm instead of BatchNorm2d
  int32_t window;           // 0 periodic Hann, 1 rectangular (ConvSTFT win_type None), 2 window_values
  int32_t pad_;
  const double* window_values;   // window == 2: win_len values (valid during build_plan only)
};

struct ParamInfo {
  std::string name;
  std::vector<int64_t> shape;
  int64_t numel, off;       // element offset inside A_PARAM (trainable) or A_STATE (buffers)
  int32_t arena;            // A_PARAM or A_STATE
};

struct BufInfo {
  int64_t off, bytes;
  int32_t dtype;
};

struct SyncPoint { int32_t phase, op; Ptr buf; int64_t count; int32_t dtype; };

struct Plan {
  ModelConfig cfg;
  int32_t T, NF;
  std::vector<ParamInfo> params;               // trainable, reference state_dict order
  std::vector<ParamInfo> state;                // BatchNorm running_mean / running_var
  std::map<std::string, BufInfo> bufs;         // named workspace buffers (tests / debugging)
  int64_t arena_bytes[A_COUNT];
  std::vector<char> consts;                    // host image of A_CONST
  std::vector<Op> fwd, bwd;
  std::vector<SyncPoint> syncs;                // SyncBN all-reduce points, in execution order per phase
  int32_t bucket_op = -1;                      // grad_buckets == 2: backward op index of the first bucket's UNPACK
  int64_t bucket_elem = 0;                     //                    first flat gradient element of that bucket
  int64_t bucket_end = -1;                     //                    one past its last element (-1: the end of the arena)
  std::string error;
};

Plan* build_dccrn_plan(const ModelConfig& cfg);
Plan* build_crn_plan(const ModelConfig& cfg);
Plan* build_plan(const ModelConfig& cfg);

}  // namespace sefd
