// translation unit: JointDiBS + LinearGaussian kernels and their launchers (kernels_joint.h)
#define DIBS_TU_LIN
#include "kernels_joint.h"
