/* hashgrid.h -- string-keyed chained hash table memoising node values (mirrors src/hashgrid.h:44-50). */
#ifndef C3SC_HASHTABLE_H
#define C3SC_HASHTABLE_H
#include <stddef.h>

char *size_t_a_to_char(size_t *arr, size_t n, char *buffer /* >= 256 bytes */); /* hashgrid.c:49-61 */
size_t c3sc_hashchar(size_t size, const char *str);                            /* hashgrid.c:75-87 (static there) */

struct HTable;
struct HTable *htable_create(size_t size);
void htable_destroy(struct HTable *);
int htable_add_element(struct HTable *, char *key, double *data, size_t N);     /* never checks duplicates */
double *htable_get_element(struct HTable *, char *key, size_t *N);
#endif
