// TEST INFRASTRUCTURE: see context_gpu.h in this directory tree
#pragma once
#include "caffe2/core/context_gpu.h"
