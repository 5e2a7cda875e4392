// xsum_model.cpp -- host model of the order-exact parallel summation (csrc/xsum.h, csrc/xsum.hip).
// TEST INFRASTRUCTURE ONLY: it runs the same arithmetic (xs_element / xs_join / xs_applies) and the same
// hierarchy (cells of 128, groups of 64 cells, drop to the first failing group -> cell -> elements) on the CPU,
// so tests/test_xsum_model.py can compare it with the plain left-to-right sum on adversarial inputs without a GPU.
#include <vector>

#include "../conjugate_gradient_amd/csrc/xsum.h"

namespace {

double tree_sum(const double *t, int64_t n)  // any approximate sum will do for the prediction
{
    if (n <= 0) return 0.0;
    if (n == 1) return t[0];
    const int64_t h = n / 2;
    return tree_sum(t, h) + tree_sum(t + h, n - h);
}

xs_summary summarize_cell(const double *t, int64_t cnt, double predicted_start)
{
    int32_t E, neg;
    bool    zeros = true;  // a cell of zeros only is the identity whatever the prediction (xsum.hip, xs_summarize)
    for (int64_t k = 0; k < cnt; ++k) zeros = zeros && t[k] == 0.0;
    if (zeros) return xs_identity();
    if (!xs_binade_of(predicted_start, &E, &neg)) return xs_regular(0, 0, 0, 0, 0, true);
    // plain run, or (if the predicted mantissa would leave the binade upwards inside the cell) a run with one crossing
    return xs_cell_summary(t, (int)cnt, E, neg, xs_mantissa(predicted_start));
}

} // namespace

extern "C" {

// stats[0] groups applied whole, [1] cells applied whole, [2] cells added element by element, [3] applied summaries that
// held binade crossings
double xsum_model_sum(const double *t, int64_t n, int64_t *stats)
{
    const int64_t ncells = (n + XS_CELL - 1) / XS_CELL, ngroups = (ncells + XS_GROUP - 1) / XS_GROUP;
    std::vector<double>     cellsum(ncells);
    std::vector<xs_summary> cs(ncells), gs(ngroups);
    for (int64_t c = 0; c < ncells; ++c) {
        const int64_t b = c * XS_CELL, cnt = n - b < XS_CELL ? n - b : XS_CELL;
        cellsum[c] = tree_sum(t + b, cnt);
    }
    double prefix = 0.0;
    for (int64_t c = 0; c < ncells; ++c) {
        const int64_t b = c * XS_CELL, cnt = n - b < XS_CELL ? n - b : XS_CELL;
        cs[c] = summarize_cell(t + b, cnt, prefix);
        prefix += cellsum[c];
    }
    for (int64_t g = 0; g < ngroups; ++g) {
        xs_summary q = xs_identity();
        for (int64_t c = g * XS_GROUP; c < ncells && c < (g + 1) * XS_GROUP; ++c) q = xs_join(q, cs[c]);
        gs[g] = q;
    }
    int64_t st[4] = {0, 0, 0, 0};
    double  s = 0.0;
    for (int64_t g = 0; g < ngroups; ++g) {
        int32_t E, neg;
        if (xs_binade_of(s, &E, &neg) && xs_applies(gs[g], E, neg, xs_mantissa(s))) {
            s = xs_apply(gs[g], E, neg, xs_mantissa(s));
            ++st[0];
            st[3] += (xs_K(gs[g]) > 0 || (gs[g].meta & XS_TIE)) && gs[g].E >= 0;
            continue;
        }
        for (int64_t c = g * XS_GROUP; c < ncells && c < (g + 1) * XS_GROUP; ++c) {
            if (xs_binade_of(s, &E, &neg) && xs_applies(cs[c], E, neg, xs_mantissa(s))) {
                s = xs_apply(cs[c], E, neg, xs_mantissa(s));
                ++st[1];
                st[3] += (xs_K(cs[c]) > 0 || (cs[c].meta & XS_TIE)) && cs[c].E >= 0;
                continue;
            }
            const int64_t b = c * XS_CELL, cnt = n - b < XS_CELL ? n - b : XS_CELL;
            for (int64_t k = 0; k < cnt; ++k) s = s + t[b + k];
            ++st[2];
        }
    }
    if (stats) stats[0] = st[0], stats[1] = st[1], stats[2] = st[2], stats[3] = st[3];
    return s;
}

double xsum_model_sequential(const double *t, int64_t n)
{
    double s = 0.0;
    for (int64_t k = 0; k < n; ++k) s = s + t[k];
    return s;
}

} // extern "C"
