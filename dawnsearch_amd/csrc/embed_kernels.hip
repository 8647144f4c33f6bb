// embed_kernels.hip — MiniLM-L6-v2 forward kernels (placeholder translation unit; filled in below).
#include "kernels.hpp"
