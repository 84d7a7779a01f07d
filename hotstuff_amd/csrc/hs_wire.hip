// hs_wire.hip - placeholder translation unit (wire decode + batch signing land here)
#include "engine.h"
