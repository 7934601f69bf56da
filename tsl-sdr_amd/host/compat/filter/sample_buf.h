/* compat include tree: a reference-style front end or back end (#include <multifm/receiver.h>, <filter/sample_buf.h>,
 * <config/engine.h>, <tsl/...>) compiles unchanged against the MI355X receiver when this directory comes first on the
 * include path (tests/test_host.py compiles multifm/file_if.c of the reference tree this way). */
#pragma once
#include "../../mfm_receiver.h"
