#include <hip/hip_runtime.h>
