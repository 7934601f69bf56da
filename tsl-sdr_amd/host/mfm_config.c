/*
 * mfm_config.c - recursive-descent JSON reader behind mfm_config.h.
 */
#include "mfm_config.h"

#include <ctype.h>
#include <errno.h>
#include <math.h>

enum json_kind { J_NULL, J_BOOL, J_NUM, J_STR, J_ARR, J_OBJ };

struct json_node {
    enum json_kind kind;
    double num;
    bool is_int;     /* number written without fraction / exponent */
    char *str;       /* J_STR */
    /* children: J_ARR uses items[], J_OBJ uses keys[] + items[] */
    size_t nr, cap;
    char **keys;
    struct json_node **items;
};

struct parser {
    const char *p, *end;
    const char *err;
};

static void node_free(struct json_node *n)
{
    if (!n) {
        return;
    }
    for (size_t i = 0; i < n->nr; i++) {
        if (n->keys) {
            free(n->keys[i]);
        }
        node_free(n->items[i]);
    }
    free(n->keys);
    free(n->items);
    free(n->str);
    free(n);
}

static struct json_node *node_new(enum json_kind k)
{
    struct json_node *n = calloc(1, sizeof(*n));
    if (n) {
        n->kind = k;
    }
    return n;
}

static bool node_push(struct json_node *parent, char *key, struct json_node *child)
{
    if (parent->nr == parent->cap) {
        size_t cap = parent->cap ? parent->cap * 2 : 8;
        struct json_node **it = realloc(parent->items, cap * sizeof(*it));
        if (!it) {
            return false;
        }
        parent->items = it;
        if (parent->kind == J_OBJ) {
            char **ks = realloc(parent->keys, cap * sizeof(*ks));
            if (!ks) {
                return false;
            }
            parent->keys = ks;
        }
        parent->cap = cap;
    }
    if (parent->kind == J_OBJ) {
        parent->keys[parent->nr] = key;
    }
    parent->items[parent->nr++] = child;
    return true;
}

static void skip_ws(struct parser *ps)
{
    while (ps->p < ps->end && isspace((unsigned char)*ps->p)) {
        ps->p++;
    }
}

static char *parse_string_raw(struct parser *ps)
{
    if (ps->p >= ps->end || *ps->p != '"') {
        ps->err = "expected string";
        return NULL;
    }
    ps->p++;
    size_t cap = 32, len = 0;
    char *out = malloc(cap);
    while (ps->p < ps->end && *ps->p != '"') {
        char c = *ps->p++;
        if (c == '\\' && ps->p < ps->end) {
            char e = *ps->p++;
            switch (e) {
            case 'n': c = '\n'; break;
            case 't': c = '\t'; break;
            case 'r': c = '\r'; break;
            case 'b': c = '\b'; break;
            case 'f': c = '\f'; break;
            case 'u':
                /* keep the low byte of \uXXXX; configuration strings are paths and identifiers */
                if (ps->end - ps->p >= 4) {
                    char hex[5] = { ps->p[0], ps->p[1], ps->p[2], ps->p[3], 0 };
                    c = (char)strtol(hex, NULL, 16);
                    ps->p += 4;
                }
                break;
            default: c = e; break;
            }
        }
        if (len + 2 > cap) {
            cap *= 2;
            out = realloc(out, cap);
        }
        out[len++] = c;
    }
    if (ps->p >= ps->end) {
        free(out);
        ps->err = "unterminated string";
        return NULL;
    }
    ps->p++; /* closing quote */
    out[len] = '\0';
    return out;
}

static struct json_node *parse_value(struct parser *ps, int depth)
{
    skip_ws(ps);
    if (ps->p >= ps->end || depth > 64) {
        ps->err = "unexpected end";
        return NULL;
    }
    char c = *ps->p;
    if (c == '{' || c == '[') {
        const bool obj = c == '{';
        struct json_node *n = node_new(obj ? J_OBJ : J_ARR);
        ps->p++;
        skip_ws(ps);
        if (ps->p < ps->end && *ps->p == (obj ? '}' : ']')) {
            ps->p++;
            return n;
        }
        for (;;) {
            char *key = NULL;
            skip_ws(ps);
            if (obj) {
                key = parse_string_raw(ps);
                if (!key) {
                    node_free(n);
                    return NULL;
                }
                skip_ws(ps);
                if (ps->p >= ps->end || *ps->p != ':') {
                    free(key);
                    node_free(n);
                    ps->err = "expected ':'";
                    return NULL;
                }
                ps->p++;
            }
            struct json_node *child = parse_value(ps, depth + 1);
            if (!child || !node_push(n, key, child)) {
                free(key);
                node_free(child);
                node_free(n);
                return NULL;
            }
            skip_ws(ps);
            if (ps->p < ps->end && *ps->p == ',') {
                ps->p++;
                continue;
            }
            if (ps->p < ps->end && *ps->p == (obj ? '}' : ']')) {
                ps->p++;
                return n;
            }
            node_free(n);
            ps->err = "expected ',' or closing bracket";
            return NULL;
        }
    }
    if (c == '"') {
        char *s = parse_string_raw(ps);
        if (!s) {
            return NULL;
        }
        struct json_node *n = node_new(J_STR);
        n->str = s;
        return n;
    }
    if (!strncmp(ps->p, "true", 4) || !strncmp(ps->p, "false", 5)) {
        struct json_node *n = node_new(J_BOOL);
        n->num = (*ps->p == 't');
        ps->p += (*ps->p == 't') ? 4 : 5;
        return n;
    }
    if (!strncmp(ps->p, "null", 4)) {
        ps->p += 4;
        return node_new(J_NULL);
    }
    /* number */
    char *endp = NULL;
    errno = 0;
    double v = strtod(ps->p, &endp);
    if (endp == ps->p) {
        ps->err = "unexpected character";
        return NULL;
    }
    struct json_node *n = node_new(J_NUM);
    n->num = v;
    n->is_int = true;
    for (const char *q = ps->p; q < endp; q++) {
        if (*q == '.' || *q == 'e' || *q == 'E') {
            n->is_int = false;
        }
    }
    ps->p = endp;
    return n;
}

static struct json_node *obj_find(struct json_node *o, const char *key)
{
    if (NULL == key) {
        return o; /* the node itself: how the atoms of an array ("gpuDevices": [0, 1]) are read */
    }
    if (!o || o->kind != J_OBJ) {
        return NULL;
    }
    for (size_t i = 0; i < o->nr; i++) {
        if (!strcmp(o->keys[i], key)) {
            return o->items[i];
        }
    }
    return NULL;
}

/* ------------------------------------------------------------------------------------- */

aresult_t config_new(struct config **pcfg)
{
    TSL_ASSERT_ARG(NULL != pcfg);
    struct config *c = calloc(1, sizeof(*c));
    if (!c) {
        return A_E_NOMEM;
    }
    c->node = node_new(J_OBJ);
    c->owner = true;
    *pcfg = c;
    return A_OK;
}

void config_delete(struct config **pcfg)
{
    if (pcfg && *pcfg) {
        if ((*pcfg)->owner) {
            node_free((*pcfg)->node);
        }
        free(*pcfg);
        *pcfg = NULL;
    }
}

aresult_t config_add_string(struct config *cfg, const char *json_text)
{
    TSL_ASSERT_ARG(NULL != cfg);
    TSL_ASSERT_ARG(NULL != json_text);
    struct parser ps = { json_text, json_text + strlen(json_text), NULL };
    struct json_node *root = parse_value(&ps, 0);
    if (!root) {
        MESSAGE("CONFIG", SEV_ERROR, "PARSE", "JSON error: %s", ps.err ? ps.err : "?");
        return A_E_INVAL;
    }
    skip_ws(&ps);
    if (ps.p != ps.end || root->kind != J_OBJ) {
        node_free(root);
        MESSAGE("CONFIG", SEV_ERROR, "PARSE", "configuration must be one JSON object");
        return A_E_INVAL;
    }
    /* merge: a key already present is replaced (multifm/multifm.c:105-111 stacks files this way) */
    for (size_t i = 0; i < root->nr; i++) {
        struct json_node *dst = cfg->node;
        bool replaced = false;
        for (size_t k = 0; k < dst->nr; k++) {
            if (!strcmp(dst->keys[k], root->keys[i])) {
                node_free(dst->items[k]);
                dst->items[k] = root->items[i];
                free(root->keys[i]);
                replaced = true;
                break;
            }
        }
        if (!replaced) {
            node_push(dst, root->keys[i], root->items[i]);
        }
        root->keys[i] = NULL;
        root->items[i] = NULL;
    }
    root->nr = 0;
    node_free(root);
    return A_OK;
}

aresult_t config_add(struct config *cfg, const char *filename)
{
    TSL_ASSERT_ARG(NULL != cfg);
    TSL_ASSERT_ARG(NULL != filename);
    FILE *fp = fopen(filename, "rb");
    if (!fp) {
        MESSAGE("CONFIG", SEV_ERROR, "OPEN", "cannot open [%s]: %s", filename, strerror(errno));
        return A_E_NOTFOUND;
    }
    fseek(fp, 0, SEEK_END);
    long sz = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    if (sz <= 0) {
        /* the reference ships empty tap files (etc/pocsag_narrow.json): nothing to merge */
        fclose(fp);
        return sz == 0 ? A_E_INVAL : A_E_INVAL;
    }
    char *text = malloc((size_t)sz + 1);
    size_t got = fread(text, 1, (size_t)sz, fp);
    fclose(fp);
    text[got] = '\0';
    aresult_t ret = config_add_string(cfg, text);
    free(text);
    return ret;
}

aresult_t config_get(struct config *cfg, struct config *sub, const char *key)
{
    TSL_ASSERT_ARG(NULL != cfg);
    TSL_ASSERT_ARG(NULL != sub);
    struct json_node *n = obj_find(cfg->node, key);
    if (!n) {
        return A_E_NOTFOUND;
    }
    sub->node = n;
    sub->owner = false;
    return A_OK;
}

aresult_t config_get_integer(struct config *cfg, int *val, const char *key)
{
    TSL_ASSERT_ARG(NULL != cfg);
    TSL_ASSERT_ARG(NULL != val);
    struct json_node *n = obj_find(cfg->node, key);
    if (!n) {
        return A_E_NOTFOUND;
    }
    if (n->kind != J_NUM || !n->is_int) {
        return A_E_INVAL;
    }
    *val = (int)n->num;
    return A_OK;
}

aresult_t config_get_float(struct config *cfg, double *val, const char *key)
{
    TSL_ASSERT_ARG(NULL != cfg);
    TSL_ASSERT_ARG(NULL != val);
    struct json_node *n = obj_find(cfg->node, key);
    if (!n) {
        return A_E_NOTFOUND;
    }
    if (n->kind != J_NUM) {
        return A_E_INVAL;
    }
    *val = n->num;
    return A_OK;
}

aresult_t config_get_string(struct config *cfg, const char **val, const char *key)
{
    TSL_ASSERT_ARG(NULL != cfg);
    TSL_ASSERT_ARG(NULL != val);
    struct json_node *n = obj_find(cfg->node, key);
    if (!n) {
        return A_E_NOTFOUND;
    }
    if (n->kind != J_STR) {
        return A_E_INVAL;
    }
    *val = n->str;
    return A_OK;
}

aresult_t config_get_boolean(struct config *cfg, bool *val, const char *key)
{
    TSL_ASSERT_ARG(NULL != cfg);
    TSL_ASSERT_ARG(NULL != val);
    struct json_node *n = obj_find(cfg->node, key);
    if (!n) {
        return A_E_NOTFOUND;
    }
    if (n->kind != J_BOOL) {
        return A_E_INVAL;
    }
    *val = n->num != 0.0;
    return A_OK;
}

aresult_t config_get_float_array(struct config *cfg, double **vals, size_t *nr_vals, const char *key)
{
    TSL_ASSERT_ARG(NULL != cfg);
    TSL_ASSERT_ARG(NULL != vals);
    TSL_ASSERT_ARG(NULL != nr_vals);
    struct json_node *n = obj_find(cfg->node, key);
    if (!n) {
        return A_E_NOTFOUND;
    }
    if (n->kind != J_ARR) {
        return A_E_INVAL;
    }
    double *out = malloc((n->nr ? n->nr : 1) * sizeof(double));
    if (!out) {
        return A_E_NOMEM;
    }
    for (size_t i = 0; i < n->nr; i++) {
        if (n->items[i]->kind != J_NUM) {
            free(out);
            return A_E_INVAL;
        }
        out[i] = n->items[i]->num;
    }
    *vals = out;
    *nr_vals = n->nr;
    return A_OK;
}

aresult_t config_array_length(struct config *arr, size_t *len)
{
    TSL_ASSERT_ARG(NULL != arr);
    TSL_ASSERT_ARG(NULL != len);
    if (!arr->node || arr->node->kind != J_ARR) {
        return A_E_INVAL;
    }
    *len = arr->node->nr;
    return A_OK;
}

aresult_t config_array_at(struct config *arr, struct config *item, size_t idx)
{
    if (!arr || !item || !arr->node || arr->node->kind != J_ARR) {
        return A_E_INVAL;
    }
    if (idx >= arr->node->nr) {
        return A_E_DONE;
    }
    item->node = arr->node->items[idx];
    item->owner = false;
    return A_OK;
}
