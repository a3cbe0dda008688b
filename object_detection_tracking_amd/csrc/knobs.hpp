// The ODT_* environment overrides: A/B and test knobs, none needed in production.
// knobs.cpp is the ONLY reader of the process environment.  The table is re-read when a handle / tracker / cosine context is
// created and at the stand-alone odt_op_* entry points (knobs_reload); plan builders and launchers ask env_knob() -- a plain
// array read, no getenv on any launch path -- and every handle remembers which variables were set when it was created:
// odt_describe lists them by name ("env_overrides") and counts them ("env_overrides_applied").
#pragma once
#include <string>
#include <vector>

namespace odt {

#define ODT_KNOB_LIST(X)                                                                                                          \
  X(CONV_SPLIT) X(CONV_SPLIT_PIPE) X(CONV_SPLIT_MINTILES) X(CONV_SPLIT3_MINTILES) X(CONV_SPLIT_MINK) X(CONV_SPLIT_MINBN)          \
  X(CONV_H2S_MAXK) X(CONV_H2_FEW_TILES) X(CONV_H2_N64) X(CONV_H2_N64_BM512) X(CONV_H2_BM64) X(CONV_H2K_SPLITK) X(CONV_H2K_FEWROWS) X(CONV_H2_BK64)    \
  X(CONV_H2_ROT) X(CONV_SPLIT3_FILLDIV) X(CONV_SPLIT3_BM) X(CONV_SPLIT3_SPLITK) X(CONV_SPLIT3_KWR) X(CONV_SPLIT3_KWR_N64)         \
  X(CONV_SPLIT3_FORCE_SPLITK) X(CONV_SPLIT_SRC2) X(CONV_SPLIT_RES2) X(SPLIT_REDUCE_BLOCKS) X(CONV_NT) X(AMAX_PER_WAVE) X(ROI_AMAX)            \
  X(CONV_CHUNK_BYTES) X(CONV_TILE) X(CONV_DEBUG) X(CONV_SMALLK) X(CONV_STAGES) X(CONV_FINE) X(CONV_TRACE)                         \
  X(FUSE_SHORTCUT) X(FUSE_RPN_HEAD) X(FUSE_BOTTLENECK) X(FUSE_ROT) X(FUSE_STEM) X(STEM_GRID) X(TAIL_OVERLAP)                      \
  X(SIDE_STREAM_PRIORITY) X(COSINE_STREAM_PRIORITY) X(TRACKER_TIMING)                                                             \
  X(EFFDET_SPLIT) X(EFFDET_FUSE_MB) X(EFFDET_FUSE_MB_MIN) X(EFFDET_WSCALE) X(EFFDET_MERGE_LEVELS) X(DW_PX) X(DW_SUMCAP) X(DW_XCD)

enum Knob : int {
#define ODT_KNOB_ENUM(n) K_##n,
  ODT_KNOB_LIST(ODT_KNOB_ENUM)
#undef ODT_KNOB_ENUM
  K_COUNT
};

struct KnobVal {
  bool set = false;      // the variable exists in the environment
  long i = 0;            // atol of its value
  double d = 0.0;        // atof
  char c0 = 0;           // first character
};

const KnobVal& env_knob(Knob k);
inline bool env_knob_off(Knob k) { const KnobVal& v = env_knob(k); return v.set && v.c0 == '0'; }     // "NAME=0"
inline long env_knob_long(Knob k, long dflt) { const KnobVal& v = env_knob(k); return v.set ? v.i : dflt; }
void knobs_reload();                              // re-read the environment (creation paths and odt_op_* only)
std::vector<std::string> knobs_active();          // "ODT_NAME=value" of every variable that is set, table order

}  // namespace odt
