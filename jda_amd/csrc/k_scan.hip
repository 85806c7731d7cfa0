// k_scan, dialect C, uniform batches (and the helpers every unit shares): see k_scan_impl.h
#define JDA_SCAN_TU_MAIN
#include "k_scan_impl.h"
