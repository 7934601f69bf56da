/* compat include tree (see multifm/receiver.h in this directory): TSL's <tsl/bits.h> (BL_CONTAINER_OF, BL_MIN2) -> mfm_tsl.h */
#pragma once
#include "../../mfm_tsl.h"
