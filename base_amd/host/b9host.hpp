// b9host.hpp -- C++ host side above the C ABI: settings, model-pack loaders, photometry reader,
// result writer and the adaptive-MCMC driver behind the singlePopMcmc / multiPopMcmc / makeCMD
// command-line surface.
//
// The reference's sources are not mounted (/root/reference/README.md:4); names, flags and file
// formats below follow the public BASE-9 conventions as [RECALL]ed and are documented in
// docs/FORMATS.md.  Nothing here computes a likelihood: every number comes from libbase9hip.so.
#pragma once
#include "../../include/base9_hip.h"

#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace b9h {

// ---- settings: "section.key" -> value, from a YAML subset plus --flag overrides -------------
class Settings {
  public:
    void load_yaml(const std::string &path);              // nested maps by indentation, scalars only
    void parse_args(int argc, char **argv);               // --key value | --key=value
    bool has(const std::string &key) const { return kv.count(key) != 0; }
    std::string str(const std::string &key, const std::string &def = "") const;
    double num(const std::string &key, double def) const;
    long integer(const std::string &key, long def) const;
    void set(const std::string &key, const std::string &value) { kv[key] = value; }
    std::string dump() const;
    // flag name ([RECALL] BASE-9 long options) -> settings key
    static const std::map<std::string, std::string> &flag_map();

  private:
    std::map<std::string, std::string> kv;
};

// ---- model pack in host memory (owns the arrays a b9_pack points to) ------------------------
struct ModelPack {
    std::vector<std::string> filters;                     // columns, in photometry order
    std::vector<double> feh, y, log_age;
    std::vector<int32_t> iso_first_eep, iso_n_eep;
    std::vector<int64_t> iso_offset;
    std::vector<double> mass, mags, abs_coeff;
    // WD cooling tracks, one per (carbonicity, mass) node in that order, each with its own age axis (ragged)
    std::vector<double> wc_carb, wc_mass, wc_log_age, wc_log_teff, wc_log_radius;   // the last three: all tracks, concatenated
    std::vector<int32_t> wc_n_age;
    std::vector<int64_t> wc_offset;
    std::vector<double> at_logg, at_log_teff, at_mags;
    int n_at_type = 0;
    int ifmr_id = B9_IFMR_WILLIAMS;
    double m_wd_up = 8.0;
    b9_pack view() const;                                 // plain-pointer view for b9_load_pack
};

// Loads <dir>/msrgb/<ms_model>.model (+ absorption.dat, wd/cooling_<wd_model>.dat,
// wd/atmos_DA.dat, wd/atmos_DB.dat when present), keeping the filter columns named in `filters`
// (in that order).  Throws std::runtime_error with a message on malformed input.
ModelPack load_model_pack(const std::string &dir, const std::string &ms_model,
                          const std::string &wd_model, const std::vector<std::string> &filters);
// The filter names a .model file provides (its %f line).
std::vector<std::string> model_filters(const std::string &dir, const std::string &ms_model);

// ---- photometry -------------------------------------------------------------------------------
struct Photometry {
    std::vector<std::string> ids, filters;
    std::vector<double> obs, sigma, mass1, mass_ratio, clust_prior;
    std::vector<int32_t> stage, wd_type, use_dbi;
    std::vector<double> filter_prior_min, filter_prior_max;
    int n_stars() const { return (int)mass1.size(); }
    b9_stars view() const;
};
// [RECALL] .phot: header "id <filters> sig<filters> mass1 massRatio stage CMprior useDBI", one
// whitespace-separated row per star; sigma < 0 marks an unused filter.  Stars outside
// [min_mag, max_mag] in filter `index` are dropped, as the reference's minMag/maxMag/index do.
Photometry read_photometry(const std::string &path, double min_mag = -1e300, double max_mag = 1e300, int index = 0);

// ---- results ------------------------------------------------------------------------------------
// [RECALL] .res: header naming the sampled columns + logPost + stage, one row per kept iteration.  [OWN] a first line
// "# <comment>" (when given) says which posterior the file holds: evaluation mode, ABI version, populations, walkers.
class ResultWriter {
  public:
    ResultWriter(const std::string &path, const std::vector<std::string> &columns, const std::string &comment = "");
    ~ResultWriter();
    void row(const std::vector<double> &values, double logpost, int stage);
  private:
    void *fp;
};

const char *param_name(int idx);          // "logAge", "Y", "FeH", "modulus", "absorption", ...

}  // namespace b9h
