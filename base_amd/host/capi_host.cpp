// capi_host.cpp -- a small C surface over the host loaders so that tests (ctypes) can check the
// parsers against the generators without a GPU.  No numerics.
#include "b9host.hpp"

#include <cstring>
#include <string>

namespace {
std::string g_err;
template <class F> int guard(F f) { try { f(); return 0; } catch (const std::exception &e) { g_err = e.what(); return -1; } }
}

extern "C" {

const char *b9h_last_error(void) { return g_err.c_str(); }

// Loads a pack and hands out a b9_pack view; the storage lives until b9h_free_pack.
int b9h_load_pack(const char *dir, const char *ms_model, const char *wd_model, const char *filters_csv,
                  void **handle, b9_pack *view)
{
    return guard([&] {
        std::vector<std::string> filters;
        std::string cur;
        for (const char *p = filters_csv; ; ++p) {
            if (*p == ',' || *p == '\0') { if (!cur.empty()) filters.push_back(cur); cur.clear(); if (!*p) break; }
            else cur += *p;
        }
        auto *pk = new b9h::ModelPack(b9h::load_model_pack(dir, ms_model, wd_model, filters));
        *handle = pk;
        *view = pk->view();
    });
}
void b9h_free_pack(void *handle) { delete static_cast<b9h::ModelPack *>(handle); }

int b9h_read_phot(const char *path, double min_mag, double max_mag, int index, void **handle, b9_stars *view,
                  char *filters_out, int filters_cap)
{
    return guard([&] {
        auto *ph = new b9h::Photometry(b9h::read_photometry(path, min_mag, max_mag, index));
        *handle = ph;
        *view = ph->view();
        std::string csv;
        for (auto &f : ph->filters) csv += (csv.empty() ? "" : ",") + f;
        std::strncpy(filters_out, csv.c_str(), (size_t)filters_cap - 1);
        filters_out[filters_cap - 1] = '\0';
    });
}
void b9h_free_phot(void *handle) { delete static_cast<b9h::Photometry *>(handle); }

// Resolves settings (YAML + flags given as one string per argv entry) and returns "key = value" lines.
int b9h_settings_dump(int argc, char **argv, char *out, int cap)
{
    return guard([&] {
        b9h::Settings st;
        st.parse_args(argc, argv);
        std::string d = st.dump();
        std::strncpy(out, d.c_str(), (size_t)cap - 1);
        out[cap - 1] = '\0';
    });
}

}  // extern "C"
