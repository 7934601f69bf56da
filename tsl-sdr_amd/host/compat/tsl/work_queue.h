/* compat include tree (see multifm/receiver.h in this directory): TSL's <tsl/work_queue.h> -> the pointer FIFO of mfm_tsl.h */
#pragma once
#include "../../mfm_tsl.h"
