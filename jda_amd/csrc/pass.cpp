// libjda.so, host side: the one translation unit that compiles Pass (one sub-batch through the device pipeline, pass.h)
// and run_device / begin_call (a call's sub-batches over its lanes, run.h) for both dialects; every other unit that
// includes those headers refers to these instantiations (extern template).
#define JDA_PASS_CPP
#include "run.h"

namespace jda {

template struct Pass<float>;
template struct Pass<double>;
JDA_RUN_INST(, float)
JDA_RUN_INST(, double)
template bool begin_call<float>(Cascador*, const PlanKey&, const ScanPlan&, int, PlanEntry**);
template bool begin_call<double>(Cascador*, const PlanKey&, const ScanPlan&, int, PlanEntry**);

}  // namespace jda
