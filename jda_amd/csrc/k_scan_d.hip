// k_scan, dialect CPP (fp64 state, round() coordinates resolved by k_prep_stage0): see k_scan_impl.h
#define JDA_SCAN_TU_DOUBLE
#include "k_scan_impl.h"
