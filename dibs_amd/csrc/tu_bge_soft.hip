// translation unit: soft-graph BGe estimator kernels and their launcher (kernels_bge_soft.h)
#define DIBS_TU_BGE_SOFT
#include "kernels_bge_soft.h"
