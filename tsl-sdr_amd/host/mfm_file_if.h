/* mfm_file_if.h - file front end constructor, as multifm/file_if.h:8. */
#pragma once

#include "mfm_receiver.h"

aresult_t file_worker_thread_new(struct receiver **pthr, struct config *cfg);
