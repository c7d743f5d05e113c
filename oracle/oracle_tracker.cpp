// placeholder translation unit; the tracker-level oracle (Sort / VisualSort predict loops) lands here.
#include "oracle.h"
