/* foam_parser.c -- readers for OpenFOAM LDU matrix dumps and volScalarField files.
 * Same contract as reference src/parser.c:29-177 (keyword scan, "<count> ( v0 v1 ... )"), written for this
 * library: one tokenizer, one generic array reader, two typed front ends. */
#include <ctype.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/parser.h"

/* next whitespace-delimited token (at most cap-1 chars kept; longer tokens are split, as "%63s" would) */
static int next_token(FILE *f, char *buf, int cap)
{
    int c;
    do c = fgetc(f); while (c != EOF && isspace(c));
    if (c == EOF) return 0;
    int len = 0;
    while (c != EOF && !isspace(c)) {
        buf[len++] = (char)c;
        if (len == cap - 1) break;
        c = fgetc(f);
    }
    buf[len] = '\0';
    return 1;
}

/* position the stream just after "<keyword> <count> (" and return count, or -1 */
static int seek_array(FILE *f, const char *keyword)
{
    char tok[64];
    while (next_token(f, tok, (int)sizeof tok)) {
        if (strcmp(tok, keyword) != 0) continue;
        int count = 0;
        if (fscanf(f, "%d", &count) != 1) {
            fprintf(stderr, "Error in parsing of the number after '%s'\n", keyword);
            return -1;
        }
        char paren = 0;
        if (fscanf(f, " %c", &paren) != 1 || paren != '(') {
            fprintf(stderr, "'(' missing after '%s'\n", keyword);
            return -1;
        }
        return count;
    }
    return -1;
}

double *parseDoubleArray(FILE *file, const char *target_word, int *out_count)
{
    const int count = seek_array(file, target_word);
    if (count < 0) return NULL;
    double *v = malloc((size_t)(count > 0 ? count : 1) * sizeof *v);
    if (!v) {
        perror("Allocation failed");
        return NULL;
    }
    for (int k = 0; k < count; ++k)
        if (fscanf(file, "%lf", &v[k]) != 1) {
            fprintf(stderr, "Error in the reading of the %d-th values of '%s'\n", k, target_word);
            free(v);
            return NULL;
        }
    if (out_count) *out_count = count;
    return v;
}

int *parseIntArray(FILE *file, const char *target_word, int *out_count)
{
    const int count = seek_array(file, target_word);
    if (count < 0) return NULL;
    int *v = malloc((size_t)(count > 0 ? count : 1) * sizeof *v);
    if (!v) {
        perror("Allocation failed");
        return NULL;
    }
    for (int k = 0; k < count; ++k)
        if (fscanf(file, "%d", &v[k]) != 1) {
            fprintf(stderr, "Error in the reading of the %d-th values of '%s'\n", k, target_word);
            free(v);
            return NULL;
        }
    if (out_count) *out_count = count;
    return v;
}

double *parseDoubleArraySolution(FILE *file, const char *target_word, int *out_count)
{
    char line[256];
    while (fgets(line, sizeof line, file)) {
        if (!strstr(line, target_word)) continue;
        int count = 0;
        if (!fgets(line, sizeof line, file) || sscanf(line, "%d", &count) != 1) {
            fprintf(stderr, "Error parsing count after '%s'\n", target_word);
            return NULL;
        }
        do {
            if (!fgets(line, sizeof line, file)) {
                fprintf(stderr, "Unexpected end of file, '(' not found\n");
                return NULL;
            }
        } while (!strchr(line, '('));
        double *v = malloc((size_t)(count > 0 ? count : 1) * sizeof *v);
        if (!v) {
            perror("Allocation failed");
            return NULL;
        }
        int got = 0;
        while (got < count && fgets(line, sizeof line, file)) {
            if (sscanf(line, "%lf", &v[got]) != 1) {
                fprintf(stderr, "Error parsing value number %d\n", got);
                free(v);
                return NULL;
            }
            ++got;
        }
        if (got != count) {
            fprintf(stderr, "Error reading values of '%s' (read %d, expected %d)\n", target_word, got, count);
            free(v);
            return NULL;
        }
        do {
            if (!fgets(line, sizeof line, file)) {
                fprintf(stderr, "Unexpected end of file, ')' not found\n");
                free(v);
                return NULL;
            }
        } while (!strchr(line, ')'));
        if (out_count) *out_count = count;
        return v;
    }
    return NULL;
}
