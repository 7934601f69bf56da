/* compat include tree (see multifm/receiver.h in this directory): TSL's <app/app.h> (app_init, app_sigint_catch,
 * app_running) -> mfm_tsl.h.  multifm/multifm.c:164 calls sleep() with no include of its own for it: TSL's header brings
 * <unistd.h> along, so this one does too. */
#pragma once
#include <unistd.h>

#include "../../mfm_tsl.h"
