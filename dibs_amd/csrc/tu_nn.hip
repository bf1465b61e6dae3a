// translation unit: JointDiBS + DenseNonlinearGaussian kernels and their launchers (kernels_nn.h)
#define DIBS_TU_NN
#include "kernels_nn.h"
