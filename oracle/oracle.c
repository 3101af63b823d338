/*
 * oracle.c -- CPU restatement of the reference's predicate path.  TEST INFRASTRUCTURE ONLY
 * (see oracle.h for provenance, pinning status and who may use it).
 *
 * Written from the reference's source; every function cites the lines it follows.  Nothing
 * here is shared with the product library: different language (C vs HIP C++), different
 * algorithm shape (one (pod,node) pair at a time on strings and sorted maps).
 */
#define _GNU_SOURCE
#include "oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------
 * Quantity text -> exact nano-units.
 * Role of kube_quantity 0.6.1 `ParsedQuantity::try_from` at src/util.rs:25-26,65,68 and
 * src/predicates.rs:29,31.  Grammar: Kubernetes apimachinery resource.Quantity
 *   <quantity> ::= <sign>? <digits>? ('.' <digits>?)? <suffix>
 *   <suffix>   ::= '' | n u m k M G T P E | Ki Mi Gi Ti Pi Ei | (e|E) <sign>? <digits>
 * Semantics here are the Kubernetes ones (binary suffixes are exact powers of 1024); see
 * SURVEY.md section 8c for where the crate is suspected to differ (outside domain D).
 * ---------------------------------------------------------------------------------------------- */
/* The parser is deliberately NOT shaped like the product's (host/quantity.cpp walks the text once and scales a mantissa by
 * a decimal exponent): here the text is first split into its five fields by a table-driven recogniser, and the value is then
 * assembled as an exact rational  digits / 10^nfrac * mul_num / mul_den  in nano-units, reduced by gcd before every
 * multiplication, so that the two share neither control flow nor arithmetic.  ("the same parser written twice".) */
typedef unsigned __int128 ora_u;

static ora_u gcd_u(ora_u a, ora_u b) {
    while (b) {
        ora_u t = a % b;
        a = b;
        b = t;
    }
    return a;
}

/* character classes of the recogniser */
enum { C_SIGN, C_DIGIT, C_DOT, C_E, C_I, C_SUF, C_END, C_BAD };
static int cls(char ch) {
    if (ch == '+' || ch == '-') return C_SIGN;
    if (ch >= '0' && ch <= '9') return C_DIGIT;
    if (ch == '.') return C_DOT;
    if (ch == 'e' || ch == 'E') return C_E;     /* exponent marker, or the decimal suffix E (10^18), or the first letter of Ei */
    if (ch == 'i') return C_I;
    if (ch == 'n' || ch == 'u' || ch == 'm' || ch == 'k' || ch == 'K' || ch == 'M' || ch == 'G' || ch == 'T' || ch == 'P') return C_SUF;
    if (ch == '\0') return C_END;
    return C_BAD;
}
/* states: 0 start, 1 after sign, 2 integer digits, 3 after '.', (3 loops on fraction digits), 4 after a suffix letter,
 * 5 after 'e'/'E', 6 after the exponent's sign, 7 exponent digits, 8 after "<letter>i", 9 accept; -1 reject.
 * A leading '.' needs digits on at least one side; that is checked on the field lengths afterwards. */
static const signed char kNext[9][8] = {
    /*            SIGN DIGIT DOT  E   I  SUF END BAD */
    /* 0 start */ { 1,   2,   3, -1, -1, -1, -1, -1},
    /* 1 sign  */ {-1,   2,   3, -1, -1, -1, -1, -1},
    /* 2 int   */ {-1,   2,   3,  5, -1,  4,  9, -1},
    /* 3 frac  */ {-1,   3,  -1,  5, -1,  4,  9, -1},
    /* 4 suf   */ {-1,  -1,  -1, -1,  8, -1,  9, -1},
    /* 5 e     */ { 6,   7,  -1, -1,  8, -1,  9, -1},   /* "1E" = 10^18, "1Ei" = 2^60, "1e3" / "1e+3" = exponent */
    /* 6 esign */ {-1,   7,  -1, -1, -1, -1, -1, -1},
    /* 7 edig  */ {-1,   7,  -1, -1, -1, -1,  9, -1},
    /* 8 bin   */ {-1,  -1,  -1, -1, -1, -1,  9, -1},
};

int ora_parse_quantity(const char *s, ora_q *out) {
    if (!s || !out) return ORA_E_PARSE;
    /* 1. recognise, remembering where the fields start */
    int state = 0, neg = 0, eneg = 0, letter = 0, binary = 0, has_exp = 0;
    const char *int0 = NULL, *frac0 = NULL, *exp0 = NULL;
    int nint = 0, nfrac = 0, nexp = 0;
    for (const char *c = s;; ++c) {
        const int k = cls(*c);
        const int nx = kNext[state][k];
        if (nx < 0) return ORA_E_PARSE;
        if (nx == 1) neg = (*c == '-');
        else if (nx == 2) { if (!nint) int0 = c; ++nint; }
        else if (nx == 3 && k == C_DIGIT) { if (!nfrac) frac0 = c; ++nfrac; }
        else if (nx == 4) { letter = *c; if (letter == 'K') letter = 0x100 | 'K'; /* K only exists as Ki */ }
        else if (nx == 5) letter = (*c == 'e') ? (0x200 | 'E') : 'E';   /* a lower-case e is only ever an exponent marker */
        else if (nx == 6) { eneg = (*c == '-'); has_exp = 1; }
        else if (nx == 7) { if (!nexp) exp0 = c; ++nexp; has_exp = 1; }
        else if (nx == 8) binary = 1;
        if (nx == 9) break;
        state = nx;
    }
    if (nint + nfrac == 0) return ORA_E_PARSE;                         /* "", ".", "+", "Ki" ... */
    if (has_exp && nexp == 0) return ORA_E_PARSE;                       /* "1e+" */
    if (!binary && (letter & 0x100)) return ORA_E_PARSE;                /* "1K" is not a suffix; "1Ki" is */
    if (!has_exp && (letter & 0x200)) return ORA_E_PARSE;               /* "1e", "1ei": neither the suffix E nor Ei */
    if (binary && (letter == 'n' || letter == 'u' || letter == 'm' || letter == 'k')) return ORA_E_PARSE;  /* "1mi", "1ki" */
    letter &= 0xFF;
    /* 2. the multiplier of the suffix as a rational in nano-units: value = digits / 10^nfrac * num / den * 10^9 */
    int e10 = 9, e2 = 0; /* 10^e10 * 2^e2 */
    if (has_exp) {
        int ev = 0;
        for (int i = 0; i < nexp; ++i) {
            ev = ev * 10 + (exp0[i] - '0');
            if (ev > 100) return ORA_E_RANGE;
        }
        e10 += eneg ? -ev : ev;
    } else if (binary) {
        const char *order = "KMGTPE";
        e2 = 10 * (int)(strchr(order, letter) - order + 1);
    } else if (letter) {
        switch (letter) {
            case 'n': e10 -= 9; break;
            case 'u': e10 -= 6; break;
            case 'm': e10 -= 3; break;
            case 'k': e10 += 3; break;
            case 'M': e10 += 6; break;
            case 'G': e10 += 9; break;
            case 'T': e10 += 12; break;
            case 'P': e10 += 15; break;
            case 'E': e10 += 18; break;
            default: return ORA_E_PARSE;
        }
    }
    e10 -= nfrac;
    /* 3. assemble: numerator = digits * 2^e2 * 10^max(e10,0), denominator = 10^max(-e10,0); exact or out of the domain */
    ora_u num = 0;
    const ora_u kMax = ((ora_u)1 << 127) - 1;
    for (int i = 0; i < nint + nfrac; ++i) {
        const int d = (i < nint ? int0[i] : frac0[i - nint]) - '0';
        if (num > (kMax - (ora_u)d) / 10) return ORA_E_RANGE;
        num = num * 10 + (ora_u)d;
    }
    ora_u den = 1;
    for (int i = 0; i < -e10; ++i) {
        if (num % 10 == 0 && num != 0) num /= 10;      /* cancel as we go: the denominator never has to hold 10^k for large k */
        else if (num == 0) break;
        else { if (den > kMax / 10) return ORA_E_RANGE; den *= 10; }
    }
    if (num != 0 && e2) {  /* 2^e2 against what is left of the denominator */
        ora_u p2 = (ora_u)1 << e2;
        const ora_u g = gcd_u(p2, den);
        p2 /= g;
        den /= g;
        if (num > kMax / p2) return ORA_E_RANGE;
        num *= p2;
    }
    for (int i = 0; i < e10 && num != 0; ++i) {
        if (num > kMax / 10) return ORA_E_RANGE;
        num *= 10;
    }
    if (num != 0) {
        const ora_u g = gcd_u(num, den);
        num /= g;
        den /= g;
    } else {
        den = 1;
    }
    if (den != 1) return ORA_E_RANGE; /* finer than a nano-unit: outside this oracle's exact domain */
    *out = neg ? -(ora_q)num : (ora_q)num;
    return ORA_OK;
}

int ora_q_to_milli(ora_q q, int64_t *out) {
    if (q % 1000000 != 0) return ORA_E_RANGE;
    ora_q m = q / 1000000;
    if (m > INT64_MAX || m < INT64_MIN) return ORA_E_RANGE;
    *out = (int64_t)m;
    return ORA_OK;
}

int ora_q_to_units(ora_q q, int64_t *out) {
    if (q % 1000000000 != 0) return ORA_E_RANGE;
    ora_q m = q / 1000000000;
    if (m > INT64_MAX || m < INT64_MIN) return ORA_E_RANGE;
    *out = (int64_t)m;
    return ORA_OK;
}

/* PodResources::new(), src/util.rs:22-29: both fields are the parse of "0" */
static int resources_new(ora_resources *r) {
    int rc = ora_parse_quantity("0", &r->cpu);
    if (rc) return rc;
    return ora_parse_quantity("0", &r->memory);
}

/* src/util.rs:54-75 total_pod_resources */
int ora_total_pod_resources(const ora_pod *pod, ora_resources *out) {
    int rc = resources_new(out); /* :55 */
    if (rc) return rc;
    if (pod->has_spec) {                                  /* :57 if let Some(spec) */
        for (uint32_t i = 0; i < pod->n_containers; ++i) { /* :58 for c in &spec.containers (only) */
            const ora_container *c = &pod->containers[i];
            if (c->has_resources && c->has_requests) { /* :59-63 */
                if (c->cpu) {                          /* :64 requests.get("cpu") */
                    ora_q q;
                    if ((rc = ora_parse_quantity(c->cpu, &q))) return rc; /* :65 expect */
                    out->cpu += q;
                }
                if (c->memory) { /* :67 */
                    ora_q q;
                    if ((rc = ora_parse_quantity(c->memory, &q))) return rc; /* :68 expect */
                    out->memory += q;
                }
            }
        }
    }
    return ORA_OK;
}

/* src/predicates.rs:21-25,34: LIST pods with field selector spec.nodeName=<node>; no phase filter */
uint32_t ora_list_pods_on_node(const ora_pod *all, uint32_t n_all, const char *node_name, const ora_pod **out,
                               uint32_t cap) {
    uint32_t n = 0;
    for (uint32_t i = 0; i < n_all; ++i) {
        if (all[i].has_spec && all[i].node_name && strcmp(all[i].node_name, node_name) == 0) {
            if (n < cap) out[n] = &all[i];
            ++n;
        }
    }
    return n;
}

/* allocatable of the node, src/predicates.rs:27-32 */
static int node_allocatable(const ora_node *node, ora_resources *avail) {
    int rc = resources_new(avail); /* :27 */
    if (rc) return rc;
    if (node->has_status && node->has_allocatable) { /* :28 */
        if (!node->alloc_cpu) return ORA_E_MISSING_KEY;                               /* :29 allocatable["cpu"] */
        if ((rc = ora_parse_quantity(node->alloc_cpu, &avail->cpu))) return rc;       /* :29 expect */
        if (!node->alloc_memory) return ORA_E_MISSING_KEY;                            /* :31 */
        if ((rc = ora_parse_quantity(node->alloc_memory, &avail->memory))) return rc; /* :31 expect */
    }
    return ORA_OK;
}

/* src/predicates.rs:20-43 can_pod_fit, LIST result injected */
int ora_can_pod_fit(const ora_pod *pod, const ora_node *node, const ora_pod *const *pods_on_node, uint32_t n_on_node,
                    int *fit) {
    ora_resources avail;
    int rc = node_allocatable(node, &avail);
    if (rc) return rc;
    for (uint32_t i = 0; i < n_on_node; ++i) { /* :36-38 */
        ora_resources r;
        if ((rc = ora_total_pod_resources(pods_on_node[i], &r))) return rc;
        avail.cpu -= r.cpu; /* SubAssign, src/util.rs:31-36 */
        avail.memory -= r.memory;
    }
    ora_resources req;
    if ((rc = ora_total_pod_resources(pod, &req))) return rc;      /* :40 */
    *fit = (req.cpu <= avail.cpu) && (req.memory <= avail.memory); /* :42 */
    return ORA_OK;
}

/* BTreeMap::get on a key-sorted array */
static const char *map_get(const ora_kv *m, uint32_t n, const char *key) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = lo + (hi - lo) / 2;
        int c = strcmp(m[mid].key, key);
        if (c == 0) return m[mid].val;
        if (c < 0) lo = mid + 1; else hi = mid;
    }
    return NULL;
}

/* src/predicates.rs:45-61 does_node_selector_match */
int ora_does_node_selector_match(const ora_pod *pod, const ora_node *node) {
    int matches = 1;                                /* :46 */
    if (pod->has_spec && pod->has_node_selector) { /* :47 */
        for (uint32_t i = 0; i < pod->n_sel; ++i) { /* :48 BTreeMap order */
            if (node->has_labels) {                 /* :49 */
                const char *v = map_get(node->labels, node->n_labels, pod->sel[i].key);
                if (!v || strcmp(v, pod->sel[i].val) != 0) { /* :50 labels.get(pk) != Some(pv) */
                    matches = 0;
                    break;
                }
            } else { /* :54-57 */
                matches = 0;
                break;
            }
        }
    }
    return matches; /* :60 */
}

/* src/predicates.rs:63-77 check_node_validity: fit first, then the selector */
int ora_check_node_validity(const ora_pod *pod, const ora_node *node, const ora_pod *const *pods_on_node,
                            uint32_t n_on_node) {
    int fit = 0;
    int rc = ora_can_pod_fit(pod, node, pods_on_node, n_on_node, &fit);
    if (rc) return rc;
    if (!fit) return ORA_REASON_NOT_ENOUGH_RESOURCES;                                      /* :68-70 */
    if (!ora_does_node_selector_match(pod, node)) return ORA_REASON_NODE_SELECTOR_MISMATCH; /* :72-74 */
    return ORA_REASON_OK;                                                                  /* :76 */
}

/* src/main.rs:51-71 select_node_for_pod, draws injected */
int ora_select_node_for_pod(const ora_pod *pod, const ora_node *nodes, uint32_t n_nodes, const ora_pod *all_pods,
                            uint32_t n_all, const uint32_t *samples, uint32_t attempts) {
    int chosen = -1;                             /* :52 node = None */
    for (uint32_t i = 0; i < attempts; ++i) {    /* :53 for _ in 0..ATTEMPTS */
        if (n_nodes == 0) continue;              /* :56 choose() on an empty slice -> None */
        if (samples[i] >= n_nodes) continue;     /* cannot happen in the reference; ABI: infeasible draw */
        const ora_node *cand = &nodes[samples[i]]; /* :56-57 */
        uint32_t cnt = ora_list_pods_on_node(all_pods, n_all, cand->name, NULL, 0);
        const ora_pod **on = (const ora_pod **)malloc(sizeof(*on) * (cnt ? cnt : 1));
        if (!on) return ORA_E_RANGE - 16;
        ora_list_pods_on_node(all_pods, n_all, cand->name, on, cnt);
        int r = ora_check_node_validity(pod, cand, on, cnt); /* :61 */
        free(on);
        if (r < 0) return r - 16; /* reference would have panicked */
        if (r == ORA_REASON_OK) { /* :63-66 */
            chosen = (int)samples[i];
            break;
        }
        /* :62 warn!(...) and try again */
    }
    return chosen; /* :70 */
}

/* ---- extension E2: taints / tolerations (DESIGN.md "Extensions") ---------------------------------
 * Kubernetes v1 ToleratesTaint: a toleration matches a taint when
 *   (toleration.effect is empty or equals taint.effect) and
 *   (toleration.key is empty with operator Exists, or keys are equal) and
 *   (operator Exists, or operator Equal/empty and values are equal).
 * Only NoSchedule and NoExecute taints filter; PreferNoSchedule never does. */
static int str_eq(const char *a, const char *b) { return strcmp(a ? a : "", b ? b : "") == 0; }
static int is_empty(const char *a) { return !a || !*a; }

static int toleration_matches(const ora_toleration *t, const ora_taint *x) {
    if (!is_empty(t->effect) && !str_eq(t->effect, x->effect)) return 0;
    const int exists = t->op && strcmp(t->op, "Exists") == 0;
    const int equal = is_empty(t->op) || strcmp(t->op, "Equal") == 0;
    if (is_empty(t->key)) return exists; /* empty key + Exists tolerates everything */
    if (!str_eq(t->key, x->key)) return 0;
    if (exists) return 1;
    return equal && str_eq(t->value, x->value);
}

int ora_tolerates_node_taints(const ora_pod *pod, const ora_node *node) {
    for (uint32_t i = 0; i < node->n_taints; ++i) {
        const ora_taint *x = &node->taints[i];
        if (!(str_eq(x->effect, "NoSchedule") || str_eq(x->effect, "NoExecute"))) continue;
        int ok = 0;
        if (pod->has_spec)
            for (uint32_t j = 0; j < pod->n_tol && !ok; ++j) ok = toleration_matches(&pod->tol[j], x);
        if (!ok) return 0;
    }
    return 1;
}

/* ---- batch driver on objects --------------------------------------------------------------------- */
typedef struct { const char *name; uint32_t idx; } name_idx;
static int cmp_name_idx(const void *a, const void *b) { return strcmp(((const name_idx *)a)->name, ((const name_idx *)b)->name); }

int ora_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int ora_eval_objects(const ora_pod *pods, uint32_t n_pods, const ora_node *nodes, uint32_t n_nodes,
                     const ora_pod *bound, uint32_t n_bound, uint32_t flags, uint64_t *out_feasible,
                     uint64_t *out_fit, int threads) {
    const uint32_t W = (n_nodes + 63u) / 64u;
    int rc = ORA_OK;
    /* one snapshot: available[n] = allocatable[n] - sum over the node's LIST (src/predicates.rs:27-38),
     * computed once per node instead of once per evaluation */
    ora_resources *avail = (ora_resources *)malloc(sizeof(*avail) * (n_nodes ? n_nodes : 1));
    ora_resources *req = (ora_resources *)malloc(sizeof(*req) * (n_pods ? n_pods : 1));
    name_idx *names = (name_idx *)malloc(sizeof(*names) * (n_nodes ? n_nodes : 1));
    if (!avail || !req || !names) { rc = ORA_E_RANGE; goto done; }
    for (uint32_t n = 0; n < n_nodes; ++n) {
        if ((rc = node_allocatable(&nodes[n], &avail[n]))) goto done;
        names[n].name = nodes[n].name;
        names[n].idx = n;
    }
    qsort(names, n_nodes, sizeof(*names), cmp_name_idx);
    for (uint32_t b = 0; b < n_bound; ++b) {
        if (!bound[b].has_spec || !bound[b].node_name) continue;
        name_idx key = {bound[b].node_name, 0};
        name_idx *hit = (name_idx *)bsearch(&key, names, n_nodes, sizeof(*names), cmp_name_idx);
        if (!hit) continue; /* bound to a node outside the snapshot */
        ora_resources r;
        if ((rc = ora_total_pod_resources(&bound[b], &r))) goto done;
        avail[hit->idx].cpu -= r.cpu;
        avail[hit->idx].memory -= r.memory;
    }
    for (uint32_t p = 0; p < n_pods; ++p)
        if ((rc = ora_total_pod_resources(&pods[p], &req[p]))) goto done;

#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads)
#endif
    for (uint32_t p = 0; p < n_pods; ++p) {
        uint64_t *frow = out_feasible ? out_feasible + (size_t)p * W : NULL;
        uint64_t *rrow = out_fit ? out_fit + (size_t)p * W : NULL;
        if (frow) memset(frow, 0, sizeof(uint64_t) * W);
        if (rrow) memset(rrow, 0, sizeof(uint64_t) * W);
        for (uint32_t n = 0; n < n_nodes; ++n) {
            int fit = 1;
            if (flags & ORA_FIT) fit = (req[p].cpu <= avail[n].cpu) && (req[p].memory <= avail[n].memory);
            int ok = fit; /* check_node_validity order: resources, then selector */
            if (ok && (flags & ORA_SEL)) ok = ora_does_node_selector_match(&pods[p], &nodes[n]);
            if (ok && (flags & ORA_TAINT)) ok = ora_tolerates_node_taints(&pods[p], &nodes[n]);
            if (frow && ok) frow[n >> 6] |= 1ull << (n & 63u);
            if (rrow && fit) rrow[n >> 6] |= 1ull << (n & 63u);
        }
    }
    (void)threads;
done:
    free(avail);
    free(req);
    free(names);
    return rc;
}

/* ---- encoded-level restatement ------------------------------------------------------------------- */
int ora_eval_encoded(uint32_t n, const int64_t *avail_cpu, const int64_t *avail_mem, const uint32_t *label_ids,
                     uint32_t n_keys, const uint64_t *taints, uint32_t p, const int64_t *req_cpu,
                     const int64_t *req_mem, const uint32_t *sel_ids, const uint64_t *tolerations,
                     const uint32_t *samples, uint32_t attempts, uint32_t flags, uint64_t *out_feasible,
                     uint64_t *out_fit, int32_t *out_binding, int threads) {
    const uint32_t W = (n + 63u) / 64u;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for schedule(static) num_threads(threads)
#endif
    for (uint32_t i = 0; i < p; ++i) {
        uint64_t *frow = out_feasible ? out_feasible + (size_t)i * W : NULL;
        uint64_t *rrow = out_fit ? out_fit + (size_t)i * W : NULL;
        if (frow) memset(frow, 0, sizeof(uint64_t) * W);
        if (rrow) memset(rrow, 0, sizeof(uint64_t) * W);
        /* best fit: lexicographic min of (mem residual, cpu residual, node) over feasible nodes */
        int have_best = 0;
        __int128 best_mem = 0, best_cpu = 0;
        int32_t best_node = -1;
        const uint64_t tol = tolerations ? tolerations[i] : 0ull;
        for (uint32_t j = 0; j < n; ++j) {
            int fit = 1;
            if (flags & ORA_FIT) fit = (req_cpu[i] <= avail_cpu[j]) && (req_mem[i] <= avail_mem[j]); /* src/predicates.rs:42 */
            int ok = fit;
            if (ok && (flags & ORA_SEL) && sel_ids) {
                for (uint32_t k = 0; k < n_keys && ok; ++k) {
                    const uint32_t s = sel_ids[(size_t)k * p + i];
                    if (s != 0u && s != label_ids[(size_t)k * n + j]) ok = 0; /* src/predicates.rs:50 */
                }
            }
            if (ok && (flags & ORA_TAINT) && taints && (taints[j] & ~tol) != 0ull) ok = 0;
            if (frow && ok) frow[j >> 6] |= 1ull << (j & 63u);
            if (rrow && fit) rrow[j >> 6] |= 1ull << (j & 63u);
            if (ok && (flags & ORA_PICK_BESTFIT)) {
                const __int128 rm = (__int128)avail_mem[j] - req_mem[i];
                const __int128 rcpu = (__int128)avail_cpu[j] - req_cpu[i];
                if (!have_best || rm < best_mem || (rm == best_mem && rcpu < best_cpu)) {
                    have_best = 1; best_mem = rm; best_cpu = rcpu; best_node = (int32_t)j;
                }
            }
        }
        if (out_binding && (flags & ORA_PICK_BESTFIT)) out_binding[i] = best_node;
        if (out_binding && (flags & ORA_PICK_SAMPLED)) {
            /* src/main.rs:53-66 on the encoded predicate: re-evaluate each drawn node from scratch */
            int32_t b = -1;
            for (uint32_t a = 0; a < attempts && b < 0; ++a) {
                const uint32_t j = samples[(size_t)i * attempts + a];
                if (j >= n) continue;
                int ok = 1;
                if (flags & ORA_FIT) ok = (req_cpu[i] <= avail_cpu[j]) && (req_mem[i] <= avail_mem[j]);
                if (ok && (flags & ORA_SEL) && sel_ids)
                    for (uint32_t k = 0; k < n_keys && ok; ++k) {
                        const uint32_t s = sel_ids[(size_t)k * p + i];
                        if (s != 0u && s != label_ids[(size_t)k * n + j]) ok = 0;
                    }
                if (ok && (flags & ORA_TAINT) && taints && (taints[j] & ~tol) != 0ull) ok = 0;
                if (ok) b = (int32_t)j;
            }
            out_binding[i] = b;
        }
    }
    (void)threads;
    return ORA_OK;
}
