// Run-time policy of libctrlhip: every CTRL_* environment variable the library understands, read ONCE (at the first query) into
// one table.  No dispatcher, plan or op calls getenv(): they ask policy_raw(key), an array lookup.  The table can be overridden per
// process through the test / experiment ABI (ctrl_policy_set, include/ctrl_hip.h) -- e.g. the GPU tests switch the adapter's stream
// lanes off and on inside one process -- and listed (ctrl_policy_get) so that a bench line can record what a run was taken with.
#pragma once
enum PolicyKey {
    P_GROUP = 0,               // CTRL_GROUP: grouped launches of sibling problems (0 off, 1 default, 2 tiles as if alone)
    P_CLIP_A2A,                // CTRL_CLIP_A2A=0: K|V all-gather instead of the all-to-all around the temporal transformers
    P_ADAPTER_SPLIT_TOP,       // CTRL_ADAPTER_SPLIT_TOP=1: one stream lane per top-level adapter block
    P_ADAPTER_LANES,           // CTRL_ADAPTER_LANES=n: stream lanes of the adapter forward (1 = everything on the caller's stream)
    P_QKV_ONE,                 // CTRL_QKV_ONE=0: Q|K and V projections as two launches
    P_CN_SPLIT,                // CTRL_CN_SPLIT: split [hi | lo] operands of the ControlNet convolutions (0 off, "dup" packed twice)
    P_CN_SPLIT_LEVELS,         // CTRL_CN_SPLIT_LEVELS: down-block levels whose convolutions take split operands
    P_CN_SPLIT_RESNET_LEVELS,  // CTRL_CN_SPLIT_RESNET_LEVELS: levels whose ResNet conv1 / conv2 take them too (default 1; 3 = all of CTRL_CN_SPLIT_LEVELS)
    P_SMALLCONV_MFMA,          // CTRL_SMALLCONV_MFMA=0: the conditioning embedder's 16/32-channel convolutions on the VALU kernel
    P_CN_AUX,                  // CTRL_CN_AUX=0: no auxiliary stream lane in the ControlNet forward
    P_STEP_OVERLAP,            // CTRL_STEP_OVERLAP=0: ctrl_step_forward runs the two modules back to back on one stream
    P_CHECK_FINITE,            // CTRL_CHECK_FINITE=1: range check of the fp16 activations
    P_PROF_DUMP,               // CTRL_PROF_DUMP=<file>: per-launch dump of the event profiler
    P_STREAM_F32,              // CTRL_STREAM_F32=0: fp16 residual streams
    P_ADAPTER_TOK_F16,         // CTRL_ADAPTER_TOK_F16=0|f: the adapter's spatial-transformer token stream in fp32 | forced fp16
    P_ADAPTER_H1_F16,          // CTRL_ADAPTER_H1_F16=1: conv1 -> GroupNorm intermediate of the adapter ResNets in fp16
    P_ATTN_NW4,                // CTRL_ATTN_NW4: 4-wave form of the D = 64 attention at long sequences
    P_ATTN_VARIANT,            // CTRL_ATTN_VARIANT=n: instruction-selection variant of the head_dim-64 long-sequence kernel
    P_IGEMM_ORDER,             // CTRL_IGEMM_ORDER: tile walk order (legacy | auto | m,G | n,G)
    P_IGEMM8,                  // CTRL_IGEMM8=0|force: which problems take the 8-phase wide tile
    P_SPLITK_INLAUNCH,         // CTRL_SPLITK_INLAUNCH=0: split-K always through the finish kernel
    P_IGEMM_FORCE,             // CTRL_IGEMM_FORCE=<tile>: tile experiments (row outputs only)
    P_SHORTK_PAIR,             // CTRL_SHORTK_PAIR=0
    P_SMALL_TILES,             // CTRL_SMALL_TILES=0: no 4-wave tiles for the short launches of the small-M chains
    P_GN_FUSED,                // CTRL_GN_FUSED=1: one-launch GroupNorm of small maps (measured slower: off)
    P_FF_FUSED,                // CTRL_FF_FUSED=0|1: the fused GEGLU feed-forward kernel (ffn.hip) for dim-512 token GEMM pairs
    P_MULTI_CN_LANES,          // CTRL_MULTI_CN_LANES=0: MultiControlNetModel runs its nets one after the other (read by the Python mirror)
    P_CN_BATCH_LANES,          // CTRL_CN_BATCH_LANES=1: ControlNetModel.forward splits a batch >= 8 over two stream lanes (Python mirror; measured: no gain)
    P_COUNT
};
// the variable's value as getenv() would return it (nullptr = unset), from the table: environment snapshot or ctrl_policy_set override
const char* policy_raw(PolicyKey k);
inline bool policy_is0(PolicyKey k) { const char* e = policy_raw(k); return e && e[0] == '0'; }      // "=0 switches it off"
inline bool policy_is1(PolicyKey k) { const char* e = policy_raw(k); return e && e[0] == '1'; }      // "=1 switches it on"
int policy_int(PolicyKey k, int dflt);                                                               // atoi, or dflt when unset
