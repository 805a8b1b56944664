// kmertools (GPU drop-in) - command line with the reference's `comp oligo`, `comp cgr -k` and
// `ctr` flags (kmertools/src/args.rs:70-130, 208-236; dispatcher :239-368) and `cov`
// (args.rs:132-172, :299-325).  clap conventions are kept: kebab-case long flags, the
// auto-derived short flags, `--flag=value`, `-k4`.  `min`: args.rs:172-205, :326-352.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include <map>
#include <string>
#include <vector>

#include "computers.hpp"
#include "seqio.hpp"

using namespace kthost;

namespace {

struct Spec {
    char shortf;
    const char *longf;
    bool takes_value;
};

[[noreturn]] void usage_error(const std::string &msg) {
    fprintf(stderr, "error: %s\n\nFor more information, try '--help'.\n", msg.c_str());
    exit(2);
}

// parses argv[from..] against `specs`; returns long-name -> value ("true" for switches)
std::map<std::string, std::string> parse_flags(int argc, char **argv, int from, const std::vector<Spec> &specs,
                                               const char *help) {
    std::map<std::string, std::string> out;
    auto find_long = [&](const std::string &n) -> const Spec * {
        for (const auto &s : specs)
            if (n == s.longf) return &s;
        return nullptr;
    };
    auto find_short = [&](char c) -> const Spec * {
        for (const auto &s : specs)
            if (c == s.shortf) return &s;
        return nullptr;
    };
    for (int i = from; i < argc; i++) {
        std::string a = argv[i];
        if (a == "-h" || a == "--help") {
            fputs(help, stdout);
            exit(0);
        }
        if (a.rfind("--", 0) == 0) {
            std::string name = a.substr(2), val;
            bool has_val = false;
            const size_t eq = name.find('=');
            if (eq != std::string::npos) {
                val = name.substr(eq + 1);
                name = name.substr(0, eq);
                has_val = true;
            }
            const Spec *s = find_long(name);
            if (!s) usage_error("unexpected argument '--" + name + "' found");
            if (s->takes_value) {
                if (!has_val) {
                    if (i + 1 >= argc) usage_error("a value is required for '--" + name + "' but none was supplied");
                    val = argv[++i];
                }
                out[s->longf] = val;
            } else {
                out[s->longf] = "true";
            }
        } else if (a.size() >= 2 && a[0] == '-' && a != "-") {
            for (size_t j = 1; j < a.size(); j++) {
                const Spec *s = find_short(a[j]);
                if (!s) usage_error(std::string("unexpected argument '-") + a[j] + "' found");
                if (s->takes_value) {
                    std::string val = a.substr(j + 1);
                    if (!val.empty() && val[0] == '=') val = val.substr(1);
                    if (val.empty()) {
                        if (i + 1 >= argc)
                            usage_error(std::string("a value is required for '-") + a[j] + "' but none was supplied");
                        val = argv[++i];
                    }
                    out[s->longf] = val;
                    break;
                }
                out[s->longf] = "true";
            }
        } else {
            usage_error("unexpected argument '" + a + "' found");
        }
    }
    return out;
}

uint64_t ranged(const std::map<std::string, std::string> &f, const char *name, uint64_t lo, uint64_t hi, bool required,
                uint64_t dflt, bool *present = nullptr) {
    auto it = f.find(name);
    if (present) *present = it != f.end();
    if (it == f.end()) {
        if (required) usage_error(std::string("the following required arguments were not provided:\n  --") + name);
        return dflt;
    }
    char *end = nullptr;
    const unsigned long long v = strtoull(it->second.c_str(), &end, 10);
    if (it->second.empty() || *end) usage_error("invalid value '" + it->second + "' for '--" + name + "': invalid digit found in string");
    if (v < lo || v > hi)
        usage_error("invalid value '" + it->second + "' for '--" + name + "': " + it->second + " is not in " +
                    std::to_string(lo) + "..=" + std::to_string(hi));
    return v;
}

std::string required_str(const std::map<std::string, std::string> &f, const char *name) {
    auto it = f.find(name);
    if (it == f.end()) usage_error(std::string("the following required arguments were not provided:\n  --") + name);
    return it->second;
}

const char *HELP_MAIN =
    "kmertools: DNA vectorisation\n\n"
    "k-mer based vectorisation for DNA sequences for\nmetagenomics and AI/ML applications\n"
    "(MI355X build: every subcommand's per-base work runs on the GPU)\n\n"
    "Usage: kmertools <COMMAND>\n\n"
    "Commands:\n"
    "  comp  Generate sequence composition based features\n"
    "  cov   Generates coverage histogram based on the reads\n"
    "  min   Bin reads using minimisers\n"
    "  ctr   Count k-mers\n"
    "  help  Print this message or the help of the given subcommand(s)\n\n"
    "Options:\n  -h, --help     Print help\n  -V, --version  Print version\n";

const char *HELP_OLIGO =
    "Generate oligonucleotide frequency vectors\n\n"
    "Usage: kmertools comp oligo [OPTIONS] --input <INPUT> --output <OUTPUT>\n\n"
    "Options:\n"
    "  -i, --input <INPUT>      Input file path\n"
    "  -o, --output <OUTPUT>    Output vectors path\n"
    "  -c, --counts             Disable normalisation and output raw counts\n"
    "  -k, --k-size <K_SIZE>    Set k-mer size [default: 3]\n"
    "  -r, --raw-count          Raw counts\n"
    "  -p, --preset <PRESET>    Output type to write [default: spc] [possible values: csv, tsv, spc]\n"
    "  -H, --header             Include header (with k-mer in ACGT format)\n"
    "  -t, --threads <THREADS>  Thread count for computations 0=auto [default: 0]\n"
    "      --device <DEVICE>    GPU index [default: 0]\n"
    "  -h, --help               Print help\n";

const char *HELP_CGR =
    "Generates Chaos Game Representations\n\n"
    "Usage: kmertools comp cgr [OPTIONS] --input <INPUT> --output <OUTPUT>\n\n"
    "Options:\n"
    "  -i, --input <INPUT>        Input file path\n"
    "  -o, --output <OUTPUT>      Output vectors path\n"
    "  -c, --counts               Disable normalisation and output raw counts (only with k-mer mode)\n"
    "  -k, --k-size <K_SIZE>      Set k-mer size or default to full sequence CGR\n"
    "  -v, --vec-size <VEC_SIZE>  Set vector size (output will be a square matrix with N=vecsize)\n"
    "  -t, --threads <THREADS>    Thread count for computations 0=auto [default: 0]\n"
    "      --device <DEVICE>      GPU index [default: 0]\n"
    "  -h, --help                 Print help\n";

const char *HELP_CTR =
    "Count k-mers\n\n"
    "Usage: kmertools ctr [OPTIONS] --input <INPUT> --output <OUTPUT> --k-size <K_SIZE>\n\n"
    "Options:\n"
    "  -i, --input <INPUT>      Input file path\n"
    "  -o, --output <OUTPUT>    Output directory path\n"
    "  -k, --k-size <K_SIZE>    k size for counting\n"
    "  -m, --memory <MEMORY>    Max memory in GB [default: 6] (accepted; the bound is the GPU's free HBM: a table\n"
    "                           that cannot hold every distinct k-mer is filled in several passes over the input)\n"
    "  -a, --acgt               Output ACGT instead of numeric values\n"
    "  -t, --threads <THREADS>  Thread count for computations 0=auto [default: 0]\n"
    "      --device <DEVICE>    GPU index [default: 0]\n"
    "      --devices <N>        Shard the table over N GPUs, --device .. --device + N - 1 [default: 1]\n"
    "  -h, --help               Print help\n";

const char *HELP_COV =
    "Generates coverage histogram based on the reads\n\n"
    "Usage: kmertools cov [OPTIONS] --input <INPUT> --output <OUTPUT>\n\n"
    "Options:\n"
    "  -i, --input <INPUT>          Input file path\n"
    "  -a, --alt-input <ALT_INPUT>  Input file path, for k-mer counting\n"
    "  -o, --output <OUTPUT>        Output directory path\n"
    "  -k, --k-size <K_SIZE>        K size for the coverage histogram [default: 15]\n"
    "  -p, --preset <PRESET>        Output type to write [default: spc] [possible values: csv, tsv, spc]\n"
    "  -s, --bin-size <BIN_SIZE>    Bin size for the coverage histogram [default: 16]\n"
    "  -c, --bin-count <BIN_COUNT>  Number of bins for the coverage histogram [default: 16]\n"
    "  -m, --memory <MEMORY>        Max memory in GB [default: 6] (accepted; the table lives in HBM)\n"
    "      --counts                 Disable normalisation and output raw counts\n"
    "  -t, --threads <THREADS>      Thread count for computations 0=auto [default: 0]\n"
    "      --device <DEVICE>        GPU index [default: 0]\n"
    "      --devices <N>            Shard the table over N GPUs, --device .. --device + N - 1 [default: 1]\n"
    "  -h, --help                   Print help\n";

int make_out_dir(const std::string &out) {
    // create_directory(&command.output).unwrap()  (args.rs:300, :354)
    if (mkdir(out.c_str(), 0777) != 0) {
        struct stat st;
        if (stat(out.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) {
            fprintf(stderr, "Error: unable to create directory: %s\n", out.c_str());
            return 101;  // the reference panics here
        }
    }
    return 0;
}

int cmd_cov(int argc, char **argv, int from) {
    const std::vector<Spec> specs = {{'i', "input", true},    {'a', "alt-input", true}, {'o', "output", true},
                                     {'k', "k-size", true},   {'p', "preset", true},    {'s', "bin-size", true},
                                     {'c', "bin-count", true}, {'m', "memory", true},   {0, "counts", false},
                                     {'t', "threads", true},  {0, "device", true},      {0, "devices", true}};
    const auto f = parse_flags(argc, argv, from, specs, HELP_COV);
    const std::string in = required_str(f, "input"), out = required_str(f, "output");
    const int k = (int)ranged(f, "k-size", 7, 31, false, 15);
    const uint64_t bin_size = ranged(f, "bin-size", 5, ~0ull, false, 16);
    const uint64_t bin_count = ranged(f, "bin-count", 5, ~0ull, false, 16);
    const uint64_t mem = ranged(f, "memory", 6, 128, false, 6);
    const int threads = (int)ranged(f, "threads", 0, 1 << 20, false, 0);
    std::string preset = f.count("preset") ? f.at("preset") : "spc";
    if (preset != "csv" && preset != "tsv" && preset != "spc")
        usage_error("invalid value '" + preset + "' for '--preset <PRESET>'\n  [possible values: csv, tsv, spc]");
    if (int rc = make_out_dir(out)) return rc;
    const std::string kin = f.count("alt-input") ? f.at("alt-input") : in;
    for (const std::string &p : {in, kin}) {
        if (format_from_path(p) == SeqFormat::Unknown) {  // "-" included: both inputs are read more than once
            fprintf(stderr, "Error: unsupported input extension (expected .fa/.fasta/.fna/.fq/.fastq[.gz]): %s\n", p.c_str());
            return 101;  // SeqFormat::get(...).unwrap()
        }
    }
    CovComputer cov(in, out, k, bin_size, bin_count);
    if (threads > 0) cov.set_threads(threads);
    if (f.count("alt-input")) cov.set_kmer_path(kin);
    if (f.count("counts")) cov.set_norm(false);
    cov.set_max_memory((double)mem);
    cov.set_delim(preset == "csv" ? "," : preset == "tsv" ? "\t" : " ");
    cov.set_device((int)ranged(f, "device", 0, 63, false, 0));
    cov.set_devices((int)ranged(f, "devices", 1, 64, false, 1));
    std::string e = cov.build_table();
    if (e.empty()) e = cov.compute_coverages();
    if (!e.empty()) {
        fprintf(stderr, "Error: %s\n", e.c_str());
        return 101;  // build_table().unwrap() / unwraps inside compute_coverages
    }
    return 0;
}

const char *HELP_MIN =
    "Bin reads using minimisers\n\n"
    "Usage: kmertools min [OPTIONS] --input <INPUT> --output <OUTPUT>\n\n"
    "Options:\n"
    "  -i, --input <INPUT>      Input file path\n"
    "  -o, --output <OUTPUT>    Output vectors path\n"
    "  -m, --m-size <M_SIZE>    Minimiser size [default: 10]\n"
    "  -w, --w-size <W_SIZE>    Window size\n"
    "                           \n"
    "                           0 - emits one minimiser per sequence (useful for sequencing reads)\n"
    "                           w_size must be longer than m_size [default: 0]\n"
    "  -p, --preset <PRESET>    Output type to write [default: s2m] [possible values: s2m, m2s]\n"
    "  -t, --threads <THREADS>  Thread count for computations 0=auto [default: 0]\n"
    "      --device <DEVICE>    GPU index [default: 0]\n"
    "  -h, --help               Print help\n";

int cmd_min(int argc, char **argv, int from) {
    const std::vector<Spec> specs = {{'i', "input", true},  {'o', "output", true},  {'m', "m-size", true}, {'w', "w-size", true},
                                     {'p', "preset", true}, {'t', "threads", true}, {0, "device", true}};
    const auto f = parse_flags(argc, argv, from, specs, HELP_MIN);
    const std::string in = required_str(f, "input"), out = required_str(f, "output");
    const int m = (int)ranged(f, "m-size", 7, 28, false, 10);
    const uint64_t w = ranged(f, "w-size", 0, ~0ull, false, 0);
    const int threads = (int)ranged(f, "threads", 0, 1 << 20, false, 0);
    const int device = (int)ranged(f, "device", 0, 63, false, 0);
    std::string preset = f.count("preset") ? f.at("preset") : "s2m";
    if (preset != "s2m" && preset != "m2s")
        usage_error("invalid value '" + preset + "' for '--preset <PRESET>'\n  [possible values: s2m, m2s]");
    if (w <= (uint64_t)m && w > 0) {  // args.rs:327-330 (returns normally)
        fprintf(stderr, "Window size must be longer than minimiser size!\n");
        return 0;
    }
    const std::string e = preset == "m2s" ? bin_sequences(w, m, in, out, threads, device)
                                          : seq_to_min(w, m, in, out, threads, device);
    if (!e.empty()) {
        fprintf(stderr, "Error: %s\n", e.c_str());
        return 101;  // the reference unwraps every failure in these two functions
    }
    return 0;
}

int cmd_oligo(int argc, char **argv, int from) {
    const std::vector<Spec> specs = {{'i', "input", true},    {'o', "output", true},  {'c', "counts", false},
                                     {'k', "k-size", true},   {'r', "raw-count", false}, {'p', "preset", true},
                                     {'H', "header", false},  {'t', "threads", true}, {0, "device", true}};
    const auto f = parse_flags(argc, argv, from, specs, HELP_OLIGO);
    const std::string in = required_str(f, "input"), out = required_str(f, "output");
    const int k = (int)ranged(f, "k-size", 3, 7, false, 3);
    const int threads = (int)ranged(f, "threads", 0, 1 << 20, false, 0);
    std::string preset = f.count("preset") ? f.at("preset") : "spc";
    if (preset != "csv" && preset != "tsv" && preset != "spc")
        usage_error("invalid value '" + preset + "' for '--preset <PRESET>'\n  [possible values: csv, tsv, spc]");
    OligoComputer com(in, out, k, !f.count("raw-count"));  // args.rs:243-248
    if (threads > 0) com.set_threads(threads);
    com.set_norm(!f.count("counts"));
    com.set_header(f.count("header") != 0);
    com.set_delim(preset == "csv" ? "," : preset == "tsv" ? "\t" : " ");
    com.set_device((int)ranged(f, "device", 0, 63, false, 0));
    const std::string e = com.vectorise();
    if (!e.empty()) fprintf(stderr, "Error: %s\n", e.c_str());  // args.rs:260-262 (returns normally)
    return 0;
}

int cmd_cgr(int argc, char **argv, int from) {
    const std::vector<Spec> specs = {{'i', "input", true},  {'o', "output", true},   {'c', "counts", false},
                                     {'k', "k-size", true}, {'v', "vec-size", true}, {'t', "threads", true},
                                     {0, "device", true}};
    const auto f = parse_flags(argc, argv, from, specs, HELP_CGR);
    const std::string in = required_str(f, "input"), out = required_str(f, "output");
    bool has_k = false, has_v = false;
    const int k = (int)ranged(f, "k-size", 3, 7, false, 0, &has_k);
    const uint64_t v = ranged(f, "vec-size", 0, ~0ull, false, 0, &has_v);
    const int threads = (int)ranged(f, "threads", 0, 1 << 20, false, 0);
    if (!has_k) {
        if (f.count("counts")) {  // args.rs:284-287
            fprintf(stderr, "Error: cannot use counts in whole sequence CGR!\n");
            return 0;
        }
        // args.rs:288-296: vecsize defaults to 1
        CgrComputer cgr(in, out, has_v ? v : 1);
        if (threads > 0) cgr.set_threads(threads);
        cgr.set_device((int)ranged(f, "device", 0, 63, false, 0));
        const std::string e = cgr.vectorise();
        if (e == "Bad nucleotide, unable to proceed") {
            // the reference unwraps this Err inside its worker (cgr.rs:95) and the process panics
            fprintf(stderr, "Error: %s\n", e.c_str());
            return 101;
        }
        if (!e.empty()) fprintf(stderr, "Error: %s\n", e.c_str());
        return 0;
    }
    // default vecsize = (k as f64).powf(4.0).powf(0.5) as u64 = k^2   (args.rs:266-269)
    const uint64_t vecsize = has_v ? v : (uint64_t)k * (uint64_t)k;
    OligoCgrComputer cgr(in, out, k, vecsize);
    if (threads > 0) cgr.set_threads(threads);
    cgr.set_norm(!f.count("counts"));
    cgr.set_device((int)ranged(f, "device", 0, 63, false, 0));
    const std::string e = cgr.vectorise();
    if (!e.empty()) fprintf(stderr, "Error: %s\n", e.c_str());
    return 0;
}

int cmd_ctr(int argc, char **argv, int from) {
    const std::vector<Spec> specs = {{'i', "input", true},  {'o', "output", true}, {'k', "k-size", true},
                                     {'m', "memory", true}, {'a', "acgt", false},  {'t', "threads", true},
                                     {0, "device", true},   {0, "devices", true}};
    const auto f = parse_flags(argc, argv, from, specs, HELP_CTR);
    const std::string in = required_str(f, "input"), out = required_str(f, "output");
    const int k = (int)ranged(f, "k-size", 10, 31, true, 0);
    const uint64_t mem = ranged(f, "memory", 6, 128, false, 6);
    const int threads = (int)ranged(f, "threads", 0, 1 << 20, false, 0);
    if (int rc = make_out_dir(out)) return rc;
    if (format_from_path(in) == SeqFormat::Unknown) {
        // CountComputer::new unwraps SeqFormat::get (counter/src/lib.rs:38): unknown extension - "-" included - panics
        fprintf(stderr, "Error: unsupported input extension (expected .fa/.fasta/.fna/.fq/.fastq[.gz]): %s\n", in.c_str());
        return 101;
    }
    CountComputer ctr(in, out, k);
    ctr.set_devices((int)ranged(f, "devices", 1, 64, false, 1));
    if (threads > 0) ctr.set_threads(threads);
    if (f.count("acgt")) ctr.set_acgt_output(true);
    ctr.set_max_memory((double)mem);
    ctr.set_device((int)ranged(f, "device", 0, 63, false, 0));
    std::string e = ctr.count();
    if (e.empty()) e = ctr.merge(true);
    if (!e.empty()) {
        fprintf(stderr, "Error: %s\n", e.c_str());
        return 101;  // count()/merge() unwrap in the reference
    }
    return 0;
}

// hidden: parse a file and print its records (CPU-only reader tests)
int cmd_debug_read(int argc, char **argv, int from) {
    if (from >= argc) return 2;
    SeqReader r;
    const bool sniff = from + 1 < argc && !strcmp(argv[from + 1], "sniff");
    if (!r.open(argv[from], sniff)) {
        fprintf(stderr, "Error: %s\n", r.error().c_str());
        return 1;
    }
    Batch b;
    uint64_t total = 0, count = 0;
    // KT_DEBUG_READ_PASSES=N: the records N times over ONE reader (rewind(): the passes of an out-of-core count); every
    // odd pass before the last is left half-way (a rewind in the middle of the stream, pieces parsed ahead and never taken)
    const int passes = getenv("KT_DEBUG_READ_PASSES") ? atoi(getenv("KT_DEBUG_READ_PASSES")) : 1;
    for (int pass = 0; pass < passes; pass++) {
        if (pass && !r.rewind()) break;
        const bool cut_short = (pass & 1) && pass + 1 < passes;
        if (pass) printf("#pass\t%d\n", pass);
        total = count = 0;
        for (;;) {
            const bool more = r.next_batch(b, 64, 2, true);  // tiny batches: exercises batch boundaries
            for (uint64_t i = 0; i < b.n_reads(); i++) {
                printf("%llu\t%s\t", (unsigned long long)(b.first_record + i), b.ids[i].c_str());
                fwrite(b.bases.data() + b.offsets[i], 1, b.offsets[i + 1] - b.offsets[i], stdout);
                printf("\n");
            }
            count += b.n_reads();
            total += b.bases.size();
            if (!more || (cut_short && count >= 1000)) break;
        }
        if (r.failed()) break;
    }
    if (r.failed()) {
        fprintf(stderr, "Error: %s\n", r.error().c_str());
        return 1;
    }
    printf("#records\t%llu\t%llu\n", (unsigned long long)count, (unsigned long long)total);
    uint64_t c2 = 0, t2 = 0;
    std::string err;
    if (SeqReader::seq_stats(argv[from], c2, t2, err)) printf("#seq_stats\t%llu\t%llu\n", (unsigned long long)c2, (unsigned long long)t2);
    return 0;
}

// hidden: checks the hand-written {:.6} formatter against snprintf("%.6f") (both round the exact
// binary value half-to-even) on count ratios, exact ties, and random doubles
int cmd_debug_fixed6(int argc, char **argv, int from) {
    const uint64_t n = from < argc ? strtoull(argv[from], nullptr, 10) : 1000000;
    uint64_t state = 0x9e3779b97f4a7c15ull, checked = 0, bad = 0;
    auto next = [&]() {
        state += 0x9e3779b97f4a7c15ull;
        uint64_t z = state;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    };
    auto check = [&](double x) {
        char a[FIXED6_BUF], b[FIXED6_BUF];
        const size_t la = format_fixed6(a, x);
        const int lb = snprintf(b, sizeof b, "%.6f", x);
        checked++;
        if (la != (size_t)lb || memcmp(a, b, la) != 0) {
            if (bad++ < 10) fprintf(stderr, "mismatch: %.17g -> '%.*s' vs '%s'\n", x, (int)la, a, b);
        }
    };
    const double specials[] = {0.0, 1.0, 0.5, 1e-7, 5e-7, 4.9999999999999998e-7, 5.0000000000000004e-7, 0.9999995, 0.99999949999999994,
                               1.0 / 128, 3.0 / 128, 1.0 / 64, 123456.7890125, 3999999999.9999995, 4.0e9, 1e300, -1.5, 2.5e-6, 1.5e-6,
                               0.1, 0.2, 0.3, 1.0 / 3.0, 2.0 / 3.0, 1e-300, 5e-324};
    for (double x : specials) check(x);
    for (uint64_t d = 1; d <= 1200; d++)             // every count ratio of a read with up to 1200 k-mers
        for (uint64_t c = 0; c <= d; c++) check((double)c / (double)d);
    for (int m = 1; m <= 40; m++)                     // dyadic values: the only exact ties
        for (uint64_t c = 1; c < 400; c += 2) check(ldexp((double)c, -m));
    for (uint64_t i = 0; i < n; i++) {
        const uint64_t r = next();
        check((double)(r >> 11) * 0x1p-53);                                        // uniform [0,1)
        check((double)(r >> 11) * 0x1p-53 * pow(10.0, (double)(next() % 19) - 9));   // wide range
        check((double)(next() % 2000000001ull) / 2000000.0 + 0.00000025);          // near half-way digits
    }
    printf("checked %llu mismatches %llu\n", (unsigned long long)checked, (unsigned long long)bad);
    return bad ? 1 : 0;
}

}  // namespace

int main(int argc, char **argv) {
    if (argc < 2) {
        fputs(HELP_MAIN, stderr);
        return 2;
    }
    const std::string cmd = argv[1];
    if (cmd == "-h" || cmd == "--help" || cmd == "help") {
        fputs(HELP_MAIN, stdout);
        return 0;
    }
    if (cmd == "-V" || cmd == "--version") {
        puts("kmertools 0.2.1 (MI355X drop-in)");
        return 0;
    }
    if (cmd == "comp") {
        if (argc < 3) usage_error("'kmertools comp' requires a subcommand but one was not provided\n  [subcommands: oligo, cgr, help]");
        const std::string sub = argv[2];
        if (sub == "oligo") return cmd_oligo(argc, argv, 3);
        if (sub == "cgr") return cmd_cgr(argc, argv, 3);
        usage_error("unrecognized subcommand '" + sub + "'");
    }
    if (cmd == "ctr") return cmd_ctr(argc, argv, 2);
    if (cmd == "debug-read") return cmd_debug_read(argc, argv, 2);
    if (cmd == "debug-fixed6") return cmd_debug_fixed6(argc, argv, 2);
    if (cmd == "debug-emit") {  // debug-emit <rows> <bins> <norm 0|1> <threads> <reps>
        if (argc < 7) return 2;
        const double s = debug_emit_bench(strtoull(argv[2], nullptr, 10), strtoull(argv[3], nullptr, 10), atoi(argv[4]) != 0,
                                          atoi(argv[5]), atoi(argv[6]));
        printf("best %.3f s\n", s);
        return 0;
    }
    if (cmd == "cov") return cmd_cov(argc, argv, 2);
    if (cmd == "min") return cmd_min(argc, argv, 2);
    usage_error("unrecognized subcommand '" + cmd + "'");
}
