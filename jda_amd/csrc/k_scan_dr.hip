// k_scan for ragged batches, dialect CPP (jdaDetectBatchCppRagged: the reference's fddb() loop as one job): see k_scan_impl.h
#define JDA_SCAN_TU_RAGGED_DOUBLE
#include "k_scan_impl.h"
