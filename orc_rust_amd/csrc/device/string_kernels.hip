// string_kernels.hip -- String/Binary/Decimal finishers (filled in below the integer path).
#include "rle_parse.h"
