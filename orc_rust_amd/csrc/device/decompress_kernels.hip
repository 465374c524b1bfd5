// decompress_kernels.hip -- ORC chunk block decompressors.
#include "rle_parse.h"
