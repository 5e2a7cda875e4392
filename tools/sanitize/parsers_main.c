#include <stdio.h>
#include <stdlib.h>
#include "parser.h"
#include "mm_reader.h"
int main(int argc, char **argv)
{
    const char *sys[] = {"tests/golden/cylinder_2000.system", "tests/golden/cylinder_8000.system"};
    for (int k = 0; k < 2; ++k) {
        FILE *f = fopen(sys[k], "r");
        if (!f) return 2;
        int nd = 0, nu = 0, nl = 0, na = 0, ns = 0;
        double *d = parseDoubleArray(f, "diag", &nd);
        double *u = parseDoubleArray(f, "upper", &nu);
        int *ua = parseIntArray(f, "upperAddr", &na);
        int *la = parseIntArray(f, "lowerAddr", &nl);
        double *s = parseDoubleArray(f, "source", &ns);
        int nx = 0;
        double *miss = parseDoubleArray(f, "no_such_word", &nx);
        printf("%s: diag %d upper %d upperAddr %d lowerAddr %d source %d missing %p/%d\n", sys[k], nd, nu, na, nl, ns, (void *)miss, nx);
        free(d); free(u); free(ua); free(la); free(s); free(miss);
        fclose(f);
    }
    FILE *f = fopen("tests/golden/Phi_2000", "r");
    int np = 0;
    double *phi = parseDoubleArraySolution(f, "internalField", &np);
    printf("Phi %d\n", np);
    free(phi);
    fclose(f);
    const char *mtx[] = {"general_60", "integer_30", "pattern_40", "plate_24x18", "skewed_1500", "nonexistent"};
    for (int k = 0; k < 6; ++k) {
        char path[256];
        snprintf(path, sizeof path, "tests/golden/mtx/%s.mtx", mtx[k]);
        int n = 0, uc = 0; double *dg = 0, *up = 0; int *r = 0, *c = 0; int sym = -1;
        int rc = mm_read_symmetric_split(path, &n, &uc, &dg, &up, &r, &c, &sym);
        printf("%s rc %d n %d upper %d sym %d\n", mtx[k], rc, n, uc, sym);
        mm_free(dg); mm_free(up); mm_free(r); mm_free(c);
    }
    return 0;
}
