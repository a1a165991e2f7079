// TEST INFRASTRUCTURE: the reference's .cu includes "caffe2/video/affine_nd_op.h"; forward to the
// reference's own header where it lies (-I/root/reference/caffe2_customized_ops on the compile line).
#pragma once
#include "video/affine_nd_op.h"
