/*
 * mfm_config.h - a small JSON configuration store with the access functions multifm's receiver-side
 * code uses (TSL's config engine, <config/engine.h>, is not available here; jansson neither).
 *
 * Same usage as in the reference (multifm/multifm.c:103-111, multifm/receiver.c:133-223):
 * several files are merged with config_add() (later files add / override top-level keys), values are
 * read with config_get_integer/float/string/boolean/float_array, sub-objects and arrays with
 * config_get() and CONFIG_ARRAY_FOR_EACH.  Key names are the reference's, verbatim, so its etc/ JSON files
 * load unchanged.
 */
#pragma once

#include "mfm_tsl.h"

struct json_node; /* opaque */

struct config {
    struct json_node *node; /* object, array or atom this view points at */
    bool owner;             /* true for the root created by config_new() */
};

#define CONFIG_INIT_EMPTY { NULL, false }

aresult_t config_new(struct config **pcfg);
void config_delete(struct config **pcfg);
/* parse a JSON file / a JSON text and merge its top-level object into cfg */
aresult_t config_add(struct config *cfg, const char *filename);
aresult_t config_add_string(struct config *cfg, const char *json_text);

/* key == NULL in the getters below: the value of the node `cfg` points at itself (an atom of an array) */
aresult_t config_get(struct config *cfg, struct config *sub, const char *key);
aresult_t config_get_integer(struct config *cfg, int *val, const char *key);
aresult_t config_get_float(struct config *cfg, double *val, const char *key);
aresult_t config_get_string(struct config *cfg, const char **val, const char *key);
aresult_t config_get_boolean(struct config *cfg, bool *val, const char *key);
/* allocates *vals with malloc(); the caller frees it (TFREE) */
aresult_t config_get_float_array(struct config *cfg, double **vals, size_t *nr_vals, const char *key);

aresult_t config_array_length(struct config *arr, size_t *len);
aresult_t config_array_at(struct config *arr, struct config *item, size_t idx);

/* for (each element `item` of array `arr`) ... ; `ret` ends A_OK after a complete walk */
#define CONFIG_ARRAY_FOR_EACH(item, arr, ret, ctr)                                                           \
    for ((ctr) = 0, (ret) = A_OK; A_OK == config_array_at((arr), &(item), (ctr)); (ctr)++)
