#include "hip_runtime.h"
