// jni_shim.cpp — libnative.so: the two JNI entry points of infoasys.cli.pangenes.PangeneNative
// (ig/native/pangene_native.h:16-25) over the C ABI of include/pandelos_amd.h.
//
// Drop-in for the reference's ig/native/build/libnative.so: same symbols, same objects read
// (PangeneIData.sequences / .sequenceGenome, library.cpp:196-265) and written (the 11 fields of Scores,
// library.cpp:542-603), same process-global dictionary semantics (library.cpp:73,192), same stdout lines
// the reference prints for its cost model (library.cpp:347-350,535-538) and the same k <= 0 behaviour
// (message + exit(1), library.cpp:90-93).  The JNI types come from include/pdl_jni_abi.h, so no JDK is
// needed to build; a JVM (or oracle/jni_harness, which plays one) calls in exactly as it calls the reference.
#include "../../include/pandelos_amd.h"
#include "../../include/pdl_jni_abi.h"

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

namespace {

pdl_ctx *g_ctx = nullptr;          // the reference keeps one process-global pair_info (library.cpp:73)
std::mutex g_mu;

[[noreturn]] void die(const char *what, const pdl_ctx *c) {
    fprintf(stderr, "pandelos_amd (libnative.so): %s: %s\n", what, pdl_last_error(c));
    exit(1);
}

pj_array make_int_array(pj_env env, const pj_int *p, pj_size n) {        // library.cpp:385-389
    pj_array a = PJ_CALL(env, NewIntArray)(env, n);
    PJ_CALL(env, SetIntArrayRegion)(env, a, 0, n, p);
    return a;
}
pj_array make_float_array(pj_env env, const pj_float *p, pj_size n) {    // library.cpp:391-395
    pj_array a = PJ_CALL(env, NewFloatArray)(env, n);
    PJ_CALL(env, SetFloatArrayRegion)(env, a, 0, n, p);
    return a;
}
void set_obj(pj_env env, pj_object o, pj_class cls, const char *name, const char *sig, pj_object v) {
    PJ_CALL(env, SetObjectField)(env, o, PJ_CALL(env, GetFieldID)(env, cls, name, sig), v);
}

}  // namespace

extern "C" {

void Java_infoasys_cli_pangenes_PangeneNative_preprocessSequences(pj_env env, pj_object, pj_object data, pj_int kvalue,
                                                                  pj_boolean onlyComplexity) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (kvalue <= 0) {                       // library.cpp:90-93
        printf("K value must be greater than 0.");
        exit(1);
    }
    // walk Vector<String> sequences and Vector<Integer> sequenceGenome (library.cpp:196-265), once
    pj_class dataClass = PJ_CALL(env, GetObjectClass)(env, data);
    pj_fieldID seqId = PJ_CALL(env, GetFieldID)(env, dataClass, "sequences", "Ljava/util/Vector;");
    pj_fieldID seqGen = PJ_CALL(env, GetFieldID)(env, dataClass, "sequenceGenome", "Ljava/util/Vector;");
    pj_object objSeq = PJ_CALL(env, GetObjectField)(env, data, seqId);
    pj_object objGen = PJ_CALL(env, GetObjectField)(env, data, seqGen);
    pj_class vecClass = PJ_CALL(env, GetObjectClass)(env, objSeq);
    pj_methodID elMethod = PJ_CALL(env, GetMethodID)(env, vecClass, "get", "(I)Ljava/lang/Object;");
    pj_methodID szMethod = PJ_CALL(env, GetMethodID)(env, vecClass, "size", "()I");
    const pj_int n = PJ_CALL(env, CallIntMethod)(env, objSeq, szMethod);

    std::vector<uint8_t> residues;
    std::vector<uint64_t> offsets((size_t) n + 1, 0);
    std::vector<uint32_t> genome_of((size_t) n, 0);
    pj_methodID intValue = nullptr;
    for (pj_int i = 0; i < n; i++) {
        pj_string str = PJ_CALL(env, CallObjectMethod)(env, objSeq, elMethod, i);
        pj_object gen = PJ_CALL(env, CallObjectMethod)(env, objGen, elMethod, i);
        if (!intValue) intValue = PJ_CALL(env, GetMethodID)(env, PJ_CALL(env, GetObjectClass)(env, gen), "intValue", "()I");
        genome_of[i] = (uint32_t) PJ_CALL(env, CallIntMethod)(env, gen, intValue);
        pj_boolean is_copy = 0;
        const pj_size len = PJ_CALL(env, GetStringLength)(env, str);
        const pj_char *chars = PJ_CALL(env, GetStringChars)(env, str, &is_copy);
        const size_t base = residues.size();
        residues.resize(base + (size_t) len);
        for (pj_size j = 0; j < len; j++) {
            if (chars[j] >= 256) {           // the reference indexes a 256-entry table with the UTF-16 unit (library.cpp:76)
                fprintf(stderr, "pandelos_amd (libnative.so): sequence %d holds a character above U+00FF; only Latin-1 residues are defined\n", i);
                exit(1);
            }
            residues[base + j] = (uint8_t) chars[j];
        }
        PJ_CALL(env, ReleaseStringChars)(env, str, chars);
        offsets[(size_t) i + 1] = residues.size();
    }

    if (!g_ctx) {
        g_ctx = pdl_create(nullptr);
        if (!g_ctx) die("no usable HIP device", nullptr);
        // The Java signatures leave no room for settings, so the drop-in takes them from the environment, once:
        // PANDELOS_AMD_OPTIONS="name=value,name=value" -> pdl_set_option (the library itself reads no environment).
        if (const char *opts = getenv("PANDELOS_AMD_OPTIONS")) {
            std::string all(opts);
            size_t at = 0;
            while (at < all.size()) {
                size_t end = all.find(',', at);
                if (end == std::string::npos) end = all.size();
                const std::string item = all.substr(at, end - at);
                const size_t eq = item.find('=');
                if (eq == std::string::npos || pdl_set_option(g_ctx, item.substr(0, eq).c_str(), atoll(item.c_str() + eq + 1)) != PDL_OK) {
                    fprintf(stderr, "pandelos_amd (libnative.so): bad PANDELOS_AMD_OPTIONS item '%s'\n", item.c_str());
                    exit(1);
                }
                at = end + 1;
            }
        }
    }
    pdl_cost cost;
    const int rc = pdl_preprocess(g_ctx, residues.data(), offsets.data(), genome_of.data(), (uint32_t) n, kvalue,
                                  onlyComplexity ? 1 : 0, &cost);
    if (rc != PDL_OK) die("preprocessSequences failed", g_ctx);
    if (cost.hash_fallback) printf("Hashing fallback!\n");           // library.cpp:117
    // the reference's cost report (library.cpp:347-350); its time estimate is a CPU calibration and is not reproduced
    printf("------------\nCOMPUTATIONAL COSTS: \nTotal cost: %llu lookups\nLinear ratio: %g\n------------\n\n",
           (unsigned long long) cost.total_cost, (double) cost.linear_ratio);
    fflush(stdout);
}

void Java_infoasys_cli_pangenes_PangeneNative_computeScores(pj_env env, pj_object, pj_int genome_id, pj_object out_scores,
                                                            pj_int /* step_size: ignored by the reference too, library.cpp:454 */) {
    pdl_ctx *c;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        c = g_ctx;
    }
    if (!c) { fprintf(stderr, "pandelos_amd (libnative.so): computeScores before preprocessSequences\n"); exit(1); }
    uint64_t gcost = 0;
    if (pdl_genome_cost(c, (uint32_t) genome_id, &gcost) != PDL_OK) die("genome id out of range", c);
    printf("Genome %d cost = %llu\n", genome_id, (unsigned long long) gcost);   // library.cpp:535-538
    fflush(stdout);

    pdl_scores s;
    if (pdl_compute_scores(c, (uint32_t) genome_id, &s) != PDL_OK) die("computeScores failed", c);
    const pj_size z = (pj_size) s.scoresCount;

    pj_class cls = PJ_CALL(env, GetObjectClass)(env, out_scores);
    PJ_CALL(env, SetIntField)(env, out_scores, PJ_CALL(env, GetFieldID)(env, cls, "scoresCount", "I"), z);       // library.cpp:544
    set_obj(env, out_scores, cls, "scoresMaxMappings", "[I", make_int_array(env, s.scoresMaxMappings, (pj_size) s.sequences));
    set_obj(env, out_scores, cls, "scores", "[F", make_float_array(env, s.scores, z));
    set_obj(env, out_scores, cls, "percs", "[F", make_float_array(env, s.percs, z));
    set_obj(env, out_scores, cls, "tr_percs", "[F", make_float_array(env, s.tr_percs, z));
    set_obj(env, out_scores, cls, "row", "[I", make_int_array(env, s.row, z));
    set_obj(env, out_scores, cls, "column", "[I", make_int_array(env, s.column, z));
    set_obj(env, out_scores, cls, "first_seq_genome", "[I", make_int_array(env, s.first_seq_genome, z));
    set_obj(env, out_scores, cls, "second_seq_genome", "[I", make_int_array(env, s.second_seq_genome, z));
    // float[rows][G] (library.cpp:577-590)
    pj_array rows = PJ_CALL(env, NewObjectArray)(env, (pj_size) s.rows, PJ_CALL(env, FindClass)(env, "[F"), nullptr);
    for (uint32_t r = 0; r < s.rows; r++)
        PJ_CALL(env, SetObjectArrayElement)(env, rows, (pj_size) r,
                                            make_float_array(env, s.max_genome_score + (size_t) r * s.genomes, (pj_size) s.genomes));
    set_obj(env, out_scores, cls, "max_genome_score", "[[F", rows);
    set_obj(env, out_scores, cls, "max_genome_score_col", "[F", make_float_array(env, s.max_genome_score_col, (pj_size) s.sequences));
    pdl_free_scores(&s);
}

}  // extern "C"
