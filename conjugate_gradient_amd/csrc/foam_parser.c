/* foam_parser.c -- readers for OpenFOAM LDU matrix dumps (".system") and volScalarField files ("Phi").
 *
 * Behavioural contract = reference src/parser.c:29-177, as its drivers use it (tests/tests_cg.c:1095-1099):
 *   LDU dump:   ... <keyword> <count> ( v0 v1 ... v{count-1} ) ...   -- tokens separated by any white space; the scan for
 *               the keyword starts at the stream's current position, so sections are requested in file order
 *   field file: a line containing the keyword, a line with the count, lines up to one containing '(', one value per
 *               line, lines up to one containing ')'
 * NULL on any failure (diagnostic on stderr), *out_count left untouched then; the caller frees the array.
 *
 * Structure is this library's own: one token scanner, one element loop parameterised by the element kind, and a small
 * line-driven state machine for the field file. */
#include <ctype.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/parser.h"

enum elem_kind { ELEM_F64, ELEM_I32 };

static void complain(const char *what, const char *keyword) { fprintf(stderr, "foam parser: %s (section '%s')\n", what, keyword); }

/* one white-space delimited token, truncated to 63 characters like a "%63s" conversion; 0 at end of file */
static int scan_token(FILE *f, char tok[64])
{
    int c = fgetc(f), len = 0;
    while (c != EOF && isspace(c)) c = fgetc(f);
    for (; c != EOF && !isspace(c) && len < 63; c = fgetc(f)) tok[len++] = (char)c;
    if (c != EOF && len == 63) ungetc(c, f);
    tok[len] = '\0';
    return len > 0;
}

/* leaves the stream behind "<keyword> <count> (" and returns the count; -1 if the keyword never comes or what follows
 * it is not a count and an opening parenthesis */
static long open_section(FILE *f, const char *keyword)
{
    char tok[64];
    int  found = 0;
    while (!found && scan_token(f, tok)) found = strcmp(tok, keyword) == 0;
    if (!found) return -1;
    int  count = 0;
    char opener = '\0';
    if (fscanf(f, "%d", &count) != 1) {
        complain("no element count behind the keyword", keyword);
        return -1;
    }
    if (fscanf(f, " %c", &opener) != 1 || opener != '(') {
        complain("no '(' behind the element count", keyword);
        return -1;
    }
    return count;
}

/* the `count` elements of an opened section into a fresh array of the given kind */
static void *read_section(FILE *f, const char *keyword, enum elem_kind kind, int *out_count)
{
    const long count = open_section(f, keyword);
    if (count < 0) return NULL;
    const size_t width = kind == ELEM_F64 ? sizeof(double) : sizeof(int);
    void        *data = malloc((size_t)(count ? count : 1) * width);
    if (!data) {
        complain("out of memory", keyword);
        return NULL;
    }
    for (long k = 0; k < count; ++k) {
        const int ok = kind == ELEM_F64 ? fscanf(f, "%lf", (double *)data + k) : fscanf(f, "%d", (int *)data + k);
        if (ok != 1) {
            fprintf(stderr, "foam parser: element %ld of section '%s' is not a number\n", k, keyword);
            free(data);
            return NULL;
        }
    }
    if (out_count) *out_count = (int)count;
    return data;
}

double *parseDoubleArray(FILE *file, const char *target_word, int *out_count)
{
    return (double *)read_section(file, target_word, ELEM_F64, out_count);
}

int *parseIntArray(FILE *file, const char *target_word, int *out_count)
{
    return (int *)read_section(file, target_word, ELEM_I32, out_count);
}

/* Field files are line oriented.  States: look for the keyword line -> count line -> '(' line -> values -> ')' line. */
double *parseDoubleArraySolution(FILE *file, const char *target_word, int *out_count)
{
    enum { WANT_KEYWORD, WANT_COUNT, WANT_OPEN, WANT_VALUES, WANT_CLOSE } state = WANT_KEYWORD;
    char    line[256];
    double *values = NULL;
    int     count = 0, have = 0;
    while (fgets(line, sizeof line, file)) {
        switch (state) {
        case WANT_KEYWORD:
            if (strstr(line, target_word)) state = WANT_COUNT;
            break;
        case WANT_COUNT:
            if (sscanf(line, "%d", &count) != 1) {
                complain("the line behind the keyword is not an element count", target_word);
                return NULL;
            }
            state = WANT_OPEN;
            break;
        case WANT_OPEN:
            if (!strchr(line, '(')) break;
            values = malloc((size_t)(count > 0 ? count : 1) * sizeof *values);
            if (!values) {
                complain("out of memory", target_word);
                return NULL;
            }
            state = count > 0 ? WANT_VALUES : WANT_CLOSE;
            break;
        case WANT_VALUES:
            if (sscanf(line, "%lf", &values[have]) != 1) {
                fprintf(stderr, "foam parser: value %d of field '%s' is not a number\n", have, target_word);
                free(values);
                return NULL;
            }
            if (++have == count) state = WANT_CLOSE;
            break;
        case WANT_CLOSE:
            if (!strchr(line, ')')) break;
            if (out_count) *out_count = count;
            return values;
        }
    }
    /* end of file before the field was complete */
    if (state == WANT_KEYWORD) return NULL;
    if (state == WANT_COUNT) complain("file ends behind the keyword", target_word);
    else if (state == WANT_OPEN) complain("file ends before '('", target_word);
    else if (state == WANT_VALUES) fprintf(stderr, "foam parser: field '%s' ends after %d of %d values\n", target_word, have, count);
    else complain("file ends before ')'", target_word);
    free(values);
    return NULL;
}
