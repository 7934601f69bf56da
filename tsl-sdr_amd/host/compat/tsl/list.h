/* compat include tree (see multifm/receiver.h in this directory): TSL's <tsl/list.h> -> the intrusive list of mfm_tsl.h */
#pragma once
#include "../../mfm_tsl.h"
