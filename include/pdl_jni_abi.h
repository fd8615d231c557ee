/*
 * pdl_jni_abi.h — the slice of the JNI binary interface that the PanDelos native
 * boundary uses, declared from the public JNI specification ("Interface Function
 * Table", Java SE 8 JNI spec ch. 4) so that the shim builds without a JDK.
 *
 * The reference binds its native library through JNI:
 *   ig/native/pangene_native.h:16-25   (the two exported symbols)
 *   ig/native/library.cpp:196-264, 385-395, 542-603   (every JNI call site)
 * It uses 18 functions of the JNIEnv table.  The table is a flat array of function
 * pointers whose slot numbers are fixed by the specification; this header names only
 * the slots that path touches and calls through them by index.  When a real JDK is on
 * the include path, <jni.h> can be used instead — the layouts are identical
 * (tests/test_boundary.py checks the slot numbers against a JDK header when one is
 * available).
 *
 * Nothing here is copied from a JDK header: types are the spec's Table 3-1 primitive
 * widths for the LP64 Linux ABI, slot numbers are the spec's function indices.
 */
#ifndef PDL_JNI_ABI_H
#define PDL_JNI_ABI_H

#include <stdarg.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* JNI primitive types (spec Table 3-1), LP64 Linux */
typedef uint8_t  pj_boolean;
typedef uint16_t pj_char;
typedef int32_t  pj_int;
typedef int32_t  pj_size;
typedef float    pj_float;

/* Reference types are opaque pointers on every HotSpot/OpenJ9 ABI */
typedef void *pj_object;
typedef pj_object pj_class;
typedef pj_object pj_string;
typedef pj_object pj_array;
typedef void *pj_fieldID;
typedef void *pj_methodID;

/* JNIEnv is a pointer to a pointer to the function table */
typedef void *const *pj_table;   /* table[slot] is a function pointer */
typedef pj_table *pj_env;        /* what the VM passes as JNIEnv*      */

/* Slot numbers (JNI spec, "Interface Function Table") */
enum {
    PJ_FindClass             = 6,
    PJ_GetObjectClass        = 31,
    PJ_GetMethodID           = 33,
    PJ_CallObjectMethod      = 34,
    PJ_CallObjectMethodV     = 35,
    PJ_CallIntMethod         = 49,
    PJ_CallIntMethodV        = 50,
    PJ_GetFieldID            = 94,
    PJ_GetObjectField        = 95,
    PJ_SetObjectField        = 104,
    PJ_SetIntField           = 109,
    PJ_GetStringLength       = 164,
    PJ_GetStringChars        = 165,
    PJ_ReleaseStringChars    = 166,
    PJ_NewObjectArray        = 172,
    PJ_SetObjectArrayElement = 174,
    PJ_NewIntArray           = 179,
    PJ_NewFloatArray         = 181,
    PJ_SetIntArrayRegion     = 211,
    PJ_SetFloatArrayRegion   = 213,
    PJ_TABLE_SLOTS           = 233   /* JNI 1.8: slots 0..232 */
};

/* Function-pointer shapes of the slots above */
typedef pj_class    (*pj_fn_FindClass)(pj_env, const char *);
typedef pj_class    (*pj_fn_GetObjectClass)(pj_env, pj_object);
typedef pj_methodID (*pj_fn_GetMethodID)(pj_env, pj_class, const char *, const char *);
typedef pj_object   (*pj_fn_CallObjectMethod)(pj_env, pj_object, pj_methodID, ...);
typedef pj_object   (*pj_fn_CallObjectMethodV)(pj_env, pj_object, pj_methodID, va_list);
typedef pj_int      (*pj_fn_CallIntMethod)(pj_env, pj_object, pj_methodID, ...);
typedef pj_int      (*pj_fn_CallIntMethodV)(pj_env, pj_object, pj_methodID, va_list);
typedef pj_fieldID  (*pj_fn_GetFieldID)(pj_env, pj_class, const char *, const char *);
typedef pj_object   (*pj_fn_GetObjectField)(pj_env, pj_object, pj_fieldID);
typedef void        (*pj_fn_SetObjectField)(pj_env, pj_object, pj_fieldID, pj_object);
typedef void        (*pj_fn_SetIntField)(pj_env, pj_object, pj_fieldID, pj_int);
typedef pj_size     (*pj_fn_GetStringLength)(pj_env, pj_string);
typedef const pj_char *(*pj_fn_GetStringChars)(pj_env, pj_string, pj_boolean *);
typedef void        (*pj_fn_ReleaseStringChars)(pj_env, pj_string, const pj_char *);
typedef pj_array    (*pj_fn_NewObjectArray)(pj_env, pj_size, pj_class, pj_object);
typedef void        (*pj_fn_SetObjectArrayElement)(pj_env, pj_array, pj_size, pj_object);
typedef pj_array    (*pj_fn_NewIntArray)(pj_env, pj_size);
typedef pj_array    (*pj_fn_NewFloatArray)(pj_env, pj_size);
typedef void        (*pj_fn_SetIntArrayRegion)(pj_env, pj_array, pj_size, pj_size, const pj_int *);
typedef void        (*pj_fn_SetFloatArrayRegion)(pj_env, pj_array, pj_size, pj_size, const pj_float *);

#define PJ_CALL(env, Name) ((pj_fn_##Name)((*(env))[PJ_##Name]))

/*
 * The two symbols the JVM resolves for infoasys.cli.pangenes.PangeneNative
 * (ig/native/pangene_native.h:16-25; Java side PangeneNative.java:14-15).
 */
__attribute__((visibility("default")))
void Java_infoasys_cli_pangenes_PangeneNative_preprocessSequences(
    pj_env env, pj_object self, pj_object data /* PangeneIData */,
    pj_int kvalue, pj_boolean onlyComplexity);

__attribute__((visibility("default")))
void Java_infoasys_cli_pangenes_PangeneNative_computeScores(
    pj_env env, pj_object self, pj_int genome_id,
    pj_object out_scores /* Scores */, pj_int step_size /* ignored, library.cpp:454 */);

#ifdef __cplusplus
}
#endif
#endif /* PDL_JNI_ABI_H */
