/* compat include tree (see multifm/receiver.h in this directory): TSL's <config/engine.h> -> mfm_config.h */
#pragma once
#include "../../mfm_config.h"
