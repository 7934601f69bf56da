/* compat include tree (see multifm/receiver.h in this directory): the TSL slice the receiver code uses -> mfm_tsl.h */
#pragma once
#include "../../mfm_tsl.h"
