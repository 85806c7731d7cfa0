// k_scan for ragged batches (images of different sizes in one pass, dialect C): see k_scan_impl.h
#define JDA_SCAN_TU_RAGGED
#include "k_scan_impl.h"
