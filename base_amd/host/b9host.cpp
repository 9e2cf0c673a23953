// b9host.cpp -- see b9host.hpp.  File formats: docs/FORMATS.md.
#include "b9host.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace b9h {

namespace {

std::string trim(const std::string &s)
{
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? "" : s.substr(a, b - a + 1);
}

std::string unquote(std::string v)
{
    v = trim(v);
    if (v.size() >= 2 && ((v.front() == '"' && v.back() == '"') || (v.front() == '\'' && v.back() == '\'')))
        v = v.substr(1, v.size() - 2);
    return v;
}

[[noreturn]] void fail(const std::string &msg) { throw std::runtime_error(msg); }

std::vector<std::string> split_ws(const std::string &line)
{
    std::vector<std::string> out;
    std::istringstream is(line);
    std::string t;
    while (is >> t) out.push_back(t);
    return out;
}

double to_double(const std::string &s, const std::string &what)
{
    std::string v = s;
    std::transform(v.begin(), v.end(), v.begin(), ::tolower);
    if (v == ".inf" || v == "inf" || v == "+inf") return INFINITY;
    if (v == "-.inf" || v == "-inf") return -INFINITY;
    char *end = nullptr;
    double d = std::strtod(s.c_str(), &end);
    if (end == s.c_str() || *end != '\0') fail("not a number (" + what + "): '" + s + "'");
    return d;
}

// value after "key=" inside a %s / %a / %c / %m / %g header line
double header_value(const std::string &line, const std::string &key)
{
    size_t p = line.find(key);
    if (p == std::string::npos) fail("header line lacks '" + key + "': " + line);
    p += key.size();
    while (p < line.size() && (line[p] == '=' || line[p] == ' ')) ++p;
    size_t e = p;
    while (e < line.size() && !isspace((unsigned char)line[e])) ++e;
    return to_double(line.substr(p, e - p), key);
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// Settings
// ---------------------------------------------------------------------------------------------
void Settings::load_yaml(const std::string &path)
{
    std::ifstream in(path);
    if (!in) fail("cannot open settings file " + path);
    std::vector<std::pair<int, std::string>> stack;      // (indent, key)
    std::string line;
    int lineno = 0;
    while (std::getline(in, line)) {
        ++lineno;
        size_t hash = std::string::npos;
        bool in_q = false;
        for (size_t i = 0; i < line.size(); ++i) {
            if (line[i] == '"' || line[i] == '\'') in_q = !in_q;
            if (line[i] == '#' && !in_q) { hash = i; break; }
        }
        if (hash != std::string::npos) line = line.substr(0, hash);
        if (trim(line).empty() || trim(line) == "---") continue;
        int indent = 0;
        while (indent < (int)line.size() && line[indent] == ' ') ++indent;
        size_t colon = line.find(':');
        if (colon == std::string::npos) fail(path + ":" + std::to_string(lineno) + ": expected 'key: value'");
        std::string key = trim(line.substr(0, colon)), value = unquote(line.substr(colon + 1));
        while (!stack.empty() && stack.back().first >= indent) stack.pop_back();
        if (value.empty()) { stack.emplace_back(indent, key); continue; }
        std::string full;
        for (auto &s : stack) full += s.second + ".";
        kv[full + key] = value;
    }
}

const std::map<std::string, std::string> &Settings::flag_map()
{
    static const std::map<std::string, std::string> m = {
        {"photFile", "general.files.photFile"}, {"outputFileBase", "general.files.outputFileBase"},
        {"modelDirectory", "general.files.modelDirectory"},
        {"msRgbModel", "general.main_sequence.msRgbModel"}, {"filterSet", "general.main_sequence.filterSet"},
        {"wdModel", "general.white_dwarfs.wdModel"}, {"ifmr", "general.white_dwarfs.ifmr"},
        {"M_wd_up", "general.white_dwarfs.M_wd_up"},
        {"priorFe_H", "general.cluster.priors.means.Fe_H"}, {"sigmaFe_H", "general.cluster.priors.sigmas.Fe_H"},
        {"priorDistMod", "general.cluster.priors.means.distMod"}, {"sigmaDistMod", "general.cluster.priors.sigmas.distMod"},
        {"priorAv", "general.cluster.priors.means.Av"}, {"sigmaAv", "general.cluster.priors.sigmas.Av"},
        {"priorY", "general.cluster.priors.means.Y"}, {"sigmaY", "general.cluster.priors.sigmas.Y"},
        {"priorCarbonicity", "general.cluster.priors.means.carbonicity"},
        {"sigmaCarbonicity", "general.cluster.priors.sigmas.carbonicity"},
        {"startingFe_H", "general.cluster.starting.Fe_H"}, {"startingDistMod", "general.cluster.starting.distMod"},
        {"startingAv", "general.cluster.starting.Av"}, {"startingY", "general.cluster.starting.Y"},
        {"startingCarbonicity", "general.cluster.starting.carbonicity"}, {"logAge", "general.cluster.starting.logAge"},
        {"startingYA", "multiPopMcmc.YA_start"}, {"startingYB", "multiPopMcmc.YB_start"},
        {"startingLambda", "multiPopMcmc.lambda_start"},
        {"minMag", "general.cluster.minMag"}, {"maxMag", "general.cluster.maxMag"}, {"index", "general.cluster.index"},
        {"burnIter", "singlePopMcmc.stage2IterMax"}, {"stage3Iter", "singlePopMcmc.stage3Iter"},
        {"runIter", "singlePopMcmc.runIter"}, {"thin", "singlePopMcmc.thin"},
        {"seed", "general.seed"}, {"verbose", "general.verbose"},
        {"walkers", "gpu.walkers"}, {"device", "gpu.device"}, {"block", "gpu.block"}, {"gpus", "gpu.gpus"},
        {"mode", "gpu.mode"}, {"marginalise", "gpu.marginalise"}, {"forceRanks", "gpu.forceRanks"}, {"tilesPerBlock", "gpu.tilesPerBlock"},
        {"resComment", "gpu.resComment"},
        {"margIsoIncrem", "sampleMass.margIsoIncrem"}, {"nMassRatios", "sampleMass.nMassRatios"},
    };
    return m;
}

void Settings::parse_args(int argc, char **argv)
{
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a.rfind("--", 0) != 0) fail("unexpected argument '" + a + "'");
        a = a.substr(2);
        std::string value;
        size_t eq = a.find('=');
        if (eq != std::string::npos) { value = a.substr(eq + 1); a = a.substr(0, eq); }
        else if (a == "verbose" || a == "marginalise" || a == "forceRanks" || a == "resComment") value = "1";
        else { if (i + 1 >= argc) fail("flag --" + a + " needs a value"); value = argv[++i]; }
        if (a == "config") { load_yaml(value); continue; }
        auto it = flag_map().find(a);
        if (it == flag_map().end()) fail("unknown flag --" + a);
        kv[it->second] = value;
    }
}

std::string Settings::str(const std::string &key, const std::string &def) const
{
    auto it = kv.find(key);
    return it == kv.end() ? def : it->second;
}
double Settings::num(const std::string &key, double def) const
{
    auto it = kv.find(key);
    return it == kv.end() ? def : to_double(it->second, key);
}
long Settings::integer(const std::string &key, long def) const { return (long)std::llround(num(key, (double)def)); }
std::string Settings::dump() const
{
    std::string s;
    for (auto &p : kv) s += p.first + " = " + p.second + "\n";
    return s;
}

// ---------------------------------------------------------------------------------------------
// Model packs
// ---------------------------------------------------------------------------------------------
b9_pack ModelPack::view() const
{
    b9_pack p{};
    p.n_filt = (int32_t)filters.size();
    p.n_feh = (int32_t)feh.size(); p.n_y = (int32_t)y.size(); p.n_age = (int32_t)log_age.size();
    p.feh = feh.data(); p.y = y.data(); p.log_age = log_age.data();
    p.iso_first_eep = iso_first_eep.data(); p.iso_n_eep = iso_n_eep.data(); p.iso_offset = iso_offset.data();
    p.n_points = (int64_t)mass.size();
    p.mass = mass.data(); p.mags = mags.data(); p.abs_coeff = abs_coeff.data();
    p.n_wc_carb = (int32_t)wc_carb.size(); p.n_wc_mass = (int32_t)wc_mass.size(); p.n_wc_points = (int64_t)wc_log_age.size();
    p.wc_carb = wc_carb.data(); p.wc_mass = wc_mass.data(); p.wc_log_age = wc_log_age.data();
    p.wc_n_age = wc_n_age.data(); p.wc_offset = wc_offset.data();
    p.wc_log_teff = wc_log_teff.data(); p.wc_log_radius = wc_log_radius.data();
    p.n_at_type = n_at_type; p.n_at_logg = (int32_t)at_logg.size(); p.n_at_teff = (int32_t)at_log_teff.size();
    p.at_logg = at_logg.data(); p.at_log_teff = at_log_teff.data(); p.at_mags = at_mags.data();
    p.ifmr_id = ifmr_id; p.m_wd_up = m_wd_up;
    return p;
}

namespace {

std::vector<int> select_columns(const std::vector<std::string> &have, const std::vector<std::string> &want,
                                const std::string &where)
{
    std::vector<int> cols;
    for (auto &f : want) {
        auto it = std::find(have.begin(), have.end(), f);
        if (it == have.end()) fail("filter '" + f + "' is not provided by " + where);
        cols.push_back((int)(it - have.begin()));
    }
    return cols;
}

struct RawIso { double feh, y, log_age; int first_eep; std::vector<double> mass, mags; };

void insert_axis(std::vector<double> &ax, double v)
{
    for (double a : ax) if (std::fabs(a - v) < 1e-9) return;
    ax.push_back(v);
}
int axis_index(const std::vector<double> &ax, double v)
{
    for (size_t i = 0; i < ax.size(); ++i) if (std::fabs(ax[i] - v) < 1e-9) return (int)i;
    return -1;
}

}  // namespace

std::vector<std::string> model_filters(const std::string &dir, const std::string &ms_model)
{
    std::ifstream in(dir + "/msrgb/" + ms_model + ".model");
    if (!in) fail("cannot open " + dir + "/msrgb/" + ms_model + ".model");
    std::string line;
    while (std::getline(in, line))
        if (line.rfind("%f", 0) == 0) { auto t = split_ws(line.substr(2)); return t; }
    fail("no %f line in " + ms_model + ".model");
}

ModelPack load_model_pack(const std::string &dir, const std::string &ms_model, const std::string &wd_model,
                          const std::vector<std::string> &filters)
{
    ModelPack pk;
    pk.filters = filters;
    const int nf = (int)filters.size();
    // ---- MS/RGB grid -------------------------------------------------------------------------
    {
        const std::string path = dir + "/msrgb/" + ms_model + ".model";
        std::ifstream in(path);
        if (!in) fail("cannot open " + path);
        std::vector<std::string> have;
        std::vector<int> cols;
        std::vector<RawIso> isos;
        double feh = NAN, y = NAN;
        std::string line;
        int lineno = 0;
        while (std::getline(in, line)) {
            ++lineno;
            if (line.empty() || line[0] == '#') continue;
            if (line[0] == '%') {
                if (line.size() < 2) continue;
                if (line[1] == 'f') { have = split_ws(line.substr(2)); cols = select_columns(have, filters, path); }
                else if (line[1] == 's') { feh = header_value(line, "[Fe/H]"); y = header_value(line, "Y"); }
                else if (line[1] == 'a') {
                    if (std::isnan(feh)) fail(path + ": %a before %s");
                    isos.push_back(RawIso{feh, y, header_value(line, "logAge"), -1, {}, {}});
                }
                continue;
            }
            if (isos.empty() || cols.empty()) fail(path + ":" + std::to_string(lineno) + ": data before %f/%s/%a");
            auto t = split_ws(line);
            if ((int)t.size() != (int)have.size() + 2) fail(path + ":" + std::to_string(lineno) + ": expected EEP, mass and one magnitude per filter");
            RawIso &iso = isos.back();
            int eep = (int)std::lround(to_double(t[0], "EEP"));
            if (iso.first_eep < 0) iso.first_eep = eep;
            else if (eep != iso.first_eep + (int)iso.mass.size()) fail(path + ":" + std::to_string(lineno) + ": EEPs must be consecutive");
            iso.mass.push_back(to_double(t[1], "mass"));
            for (int c : cols) iso.mags.push_back(to_double(t[2 + c], "magnitude"));
        }
        if (isos.empty()) fail(path + ": no isochrones");
        for (auto &i : isos) { insert_axis(pk.feh, i.feh); insert_axis(pk.y, i.y); insert_axis(pk.log_age, i.log_age); }
        std::sort(pk.feh.begin(), pk.feh.end()); std::sort(pk.y.begin(), pk.y.end()); std::sort(pk.log_age.begin(), pk.log_age.end());
        const size_t n_iso = pk.feh.size() * pk.y.size() * pk.log_age.size();
        if (n_iso != isos.size()) fail(path + ": the (FeH, Y, logAge) grid is not rectangular (" + std::to_string(isos.size()) + " isochrones for " + std::to_string(n_iso) + " grid nodes)");
        std::vector<const RawIso *> at(n_iso, nullptr);
        for (auto &i : isos) {
            size_t k = ((size_t)axis_index(pk.feh, i.feh) * pk.y.size() + axis_index(pk.y, i.y)) * pk.log_age.size() + axis_index(pk.log_age, i.log_age);
            if (at[k]) fail(path + ": duplicate isochrone");
            at[k] = &i;
        }
        pk.iso_first_eep.resize(n_iso); pk.iso_n_eep.resize(n_iso); pk.iso_offset.resize(n_iso);
        for (size_t k = 0; k < n_iso; ++k) {
            pk.iso_first_eep[k] = at[k]->first_eep; pk.iso_n_eep[k] = (int32_t)at[k]->mass.size(); pk.iso_offset[k] = (int64_t)pk.mass.size();
            pk.mass.insert(pk.mass.end(), at[k]->mass.begin(), at[k]->mass.end());
            pk.mags.insert(pk.mags.end(), at[k]->mags.begin(), at[k]->mags.end());
        }
    }
    // ---- absorption coefficients ---------------------------------------------------------------
    {
        std::map<std::string, double> coeff;
        std::ifstream in(dir + "/absorption.dat");
        std::string line;
        while (in && std::getline(in, line)) {
            if (line.empty() || line[0] == '#') continue;
            auto t = split_ws(line);
            if (t.size() >= 2) coeff[t[0]] = to_double(t[1], "absorption coefficient");
        }
        for (auto &f : filters) {
            if (!coeff.count(f)) fail("no absorption coefficient for filter '" + f + "' in " + dir + "/absorption.dat");
            pk.abs_coeff.push_back(coeff[f]);
        }
    }
    // ---- WD cooling ------------------------------------------------------------------------------
    if (!wd_model.empty()) {
        const std::string path = dir + "/wd/cooling_" + wd_model + ".dat";
        std::ifstream in(path);
        if (in) {
            struct Track { double carb, mass; std::vector<double> age, teff, rad; };
            std::vector<Track> tracks;
            double carb = 0.38;
            std::string line;
            while (std::getline(in, line)) {
                if (line.empty() || line[0] == '#') continue;
                if (line[0] == '%') {
                    if (line[1] == 'c') carb = header_value(line, "carbonicity");
                    else if (line[1] == 'm') tracks.push_back(Track{carb, header_value(line, "mass"), {}, {}, {}});
                    continue;
                }
                auto t = split_ws(line);
                if (t.size() != 3 || tracks.empty()) fail(path + ": expected 'logCoolAge logTeff logRadius' rows after %m");
                tracks.back().age.push_back(to_double(t[0], "age")); tracks.back().teff.push_back(to_double(t[1], "teff")); tracks.back().rad.push_back(to_double(t[2], "radius"));
            }
            if (tracks.empty()) fail(path + ": no cooling tracks");
            for (auto &t : tracks) { insert_axis(pk.wc_carb, t.carb); insert_axis(pk.wc_mass, t.mass); }
            std::sort(pk.wc_carb.begin(), pk.wc_carb.end()); std::sort(pk.wc_mass.begin(), pk.wc_mass.end());
            // every (carbonicity, mass) node needs exactly one track; each keeps ITS OWN age axis (tracks are ragged)
            const size_t nC = pk.wc_carb.size(), nM = pk.wc_mass.size();
            std::vector<const Track *> at(nC * nM, nullptr);
            for (auto &t : tracks) {
                const size_t k = (size_t)axis_index(pk.wc_carb, t.carb) * nM + axis_index(pk.wc_mass, t.mass);
                if (at[k]) fail(path + ": duplicate cooling track");
                if (t.age.size() < 2) fail(path + ": a cooling track needs at least two points");
                for (size_t i = 1; i < t.age.size(); ++i)
                    if (!(t.age[i] > t.age[i - 1])) fail(path + ": the cooling ages of a track must ascend");
                at[k] = &t;
            }
            for (size_t k = 0; k < at.size(); ++k) {
                if (!at[k]) fail(path + ": the (carbonicity, mass) grid of cooling tracks has holes");
                pk.wc_n_age.push_back((int32_t)at[k]->age.size());
                pk.wc_offset.push_back((int64_t)pk.wc_log_age.size());
                pk.wc_log_age.insert(pk.wc_log_age.end(), at[k]->age.begin(), at[k]->age.end());
                pk.wc_log_teff.insert(pk.wc_log_teff.end(), at[k]->teff.begin(), at[k]->teff.end());
                pk.wc_log_radius.insert(pk.wc_log_radius.end(), at[k]->rad.begin(), at[k]->rad.end());
            }
        }
        // ---- WD atmospheres ------------------------------------------------------------------------
        for (int type = 0; type < 2; ++type) {
            const std::string apath = dir + "/wd/atmos_" + (type ? "DB" : "DA") + ".dat";
            std::ifstream ia(apath);
            if (!ia) break;
            std::vector<std::string> have;
            std::vector<int> cols;
            std::vector<double> loggs, teff_axis, cur_teff;
            std::vector<std::vector<double>> blocks;       // per logg: rows of nf mags
            std::string line;
            while (std::getline(ia, line)) {
                if (line.empty() || line[0] == '#') continue;
                if (line[0] == '%') {
                    if (line[1] == 'f') { have = split_ws(line.substr(2)); cols = select_columns(have, filters, apath); }
                    else if (line[1] == 'g') {
                        if (!blocks.empty()) { if (teff_axis.empty()) teff_axis = cur_teff; else if (teff_axis != cur_teff) fail(apath + ": Teff axes differ between log g blocks"); }
                        loggs.push_back(header_value(line, "logg")); blocks.emplace_back(); cur_teff.clear();
                    }
                    continue;
                }
                auto t = split_ws(line);
                if (blocks.empty() || t.size() != have.size() + 1) fail(apath + ": expected 'logTeff mags...' rows after %g");
                cur_teff.push_back(to_double(t[0], "logTeff"));
                for (int c : cols) blocks.back().push_back(to_double(t[1 + c], "magnitude"));
            }
            if (teff_axis.empty()) teff_axis = cur_teff; else if (teff_axis != cur_teff) fail(apath + ": Teff axes differ between log g blocks");
            if (type == 0) { pk.at_logg = loggs; pk.at_log_teff = teff_axis; }
            else if (loggs != pk.at_logg || teff_axis != pk.at_log_teff) fail(apath + ": DA and DB tables must share their axes");
            for (auto &b : blocks) {
                if (b.size() != teff_axis.size() * (size_t)nf) fail(apath + ": ragged atmosphere block");
                pk.at_mags.insert(pk.at_mags.end(), b.begin(), b.end());
            }
            pk.n_at_type = type + 1;
        }
    }
    return pk;
}

// ---------------------------------------------------------------------------------------------
// Photometry
// ---------------------------------------------------------------------------------------------
b9_stars Photometry::view() const
{
    b9_stars s{};
    s.n_stars = n_stars(); s.n_filt = (int32_t)filters.size();
    s.obs = obs.data(); s.sigma = sigma.data(); s.mass1 = mass1.data(); s.mass_ratio = mass_ratio.data();
    s.clust_prior = clust_prior.data(); s.stage = stage.data(); s.wd_type = wd_type.data();
    s.filter_prior_min = filter_prior_min.data(); s.filter_prior_max = filter_prior_max.data();
    return s;
}

Photometry read_photometry(const std::string &path, double min_mag, double max_mag, int index)
{
    std::ifstream in(path);
    if (!in) fail("cannot open photometry file " + path);
    Photometry ph;
    std::string line;
    if (!std::getline(in, line)) fail(path + ": empty file");
    auto head = split_ws(line);
    if (head.empty() || head[0] != "id") fail(path + ": header must start with 'id'");
    size_t k = 1;
    while (k < head.size() && head[k].rfind("sig", 0) != 0) ph.filters.push_back(head[k++]);
    const size_t nf = ph.filters.size();
    if (nf == 0) fail(path + ": no filter columns");
    for (size_t f = 0; f < nf; ++f)
        if (k + f >= head.size() || head[k + f] != "sig" + ph.filters[f]) fail(path + ": expected column sig" + ph.filters[f]);
    k += nf;
    const char *rest[] = {"mass1", "massRatio", "stage", "CMprior", "useDBI"};
    for (int r = 0; r < 5; ++r)
        if (k + r >= head.size() || head[k + r] != rest[r]) fail(path + std::string(": expected column ") + rest[r]);
    const size_t ncol = 1 + 2 * nf + 5;
    if (index < 0 || index >= (int)nf) fail("magnitude-cut filter index out of range");
    int lineno = 1;
    while (std::getline(in, line)) {
        ++lineno;
        if (trim(line).empty() || line[0] == '#') continue;
        auto t = split_ws(line);
        if (t.size() < ncol) fail(path + ":" + std::to_string(lineno) + ": too few columns");
        std::vector<double> o(nf), s(nf);
        for (size_t f = 0; f < nf; ++f) { o[f] = to_double(t[1 + f], "magnitude"); s[f] = to_double(t[1 + nf + f], "sigma"); }
        int stage = (int)std::lround(to_double(t[1 + 2 * nf + 2], "stage"));
        // [RECALL] the magnitude window applies to MS/RGB stars in filter `index`; WDs are always kept
        if (stage != B9_STAGE_WD && (o[index] < min_mag || o[index] > max_mag)) continue;
        ph.ids.push_back(t[0]);
        ph.obs.insert(ph.obs.end(), o.begin(), o.end());
        ph.sigma.insert(ph.sigma.end(), s.begin(), s.end());
        ph.mass1.push_back(to_double(t[1 + 2 * nf], "mass1"));
        ph.mass_ratio.push_back(to_double(t[1 + 2 * nf + 1], "massRatio"));
        ph.stage.push_back(stage);
        ph.clust_prior.push_back(to_double(t[1 + 2 * nf + 3], "CMprior"));
        ph.use_dbi.push_back((int)std::lround(to_double(t[1 + 2 * nf + 4], "useDBI")));
        ph.wd_type.push_back(t.size() > ncol ? (int)std::lround(to_double(t[ncol], "wdType")) : 0);
    }
    if (ph.n_stars() == 0) fail(path + ": no stars inside the magnitude window");
    // field-star magnitude box = observed range per filter (over the stars in use) [RECALL filterPriorMin/Max]
    ph.filter_prior_min.assign(nf, 1e300); ph.filter_prior_max.assign(nf, -1e300);
    for (int i = 0; i < ph.n_stars(); ++i)
        for (size_t f = 0; f < nf; ++f) {
            if (ph.sigma[i * nf + f] <= 0) continue;
            ph.filter_prior_min[f] = std::min(ph.filter_prior_min[f], ph.obs[i * nf + f]);
            ph.filter_prior_max[f] = std::max(ph.filter_prior_max[f], ph.obs[i * nf + f]);
        }
    for (size_t f = 0; f < nf; ++f)
        if (!(ph.filter_prior_max[f] > ph.filter_prior_min[f])) { ph.filter_prior_min[f] = 0.0; ph.filter_prior_max[f] = 1.0; }
    return ph;
}

// ---------------------------------------------------------------------------------------------
// Results
// ---------------------------------------------------------------------------------------------
ResultWriter::ResultWriter(const std::string &path, const std::vector<std::string> &columns, const std::string &comment)
{
    FILE *f = std::fopen(path.c_str(), "w");
    if (!f) fail("cannot write " + path);
    if (!comment.empty()) std::fprintf(f, "# %s\n", comment.c_str());
    for (auto &c : columns) std::fprintf(f, "%12s ", c.c_str());
    std::fprintf(f, "%14s %5s\n", "logPost", "stage");
    fp = f;
}
ResultWriter::~ResultWriter() { if (fp) std::fclose((FILE *)fp); }
void ResultWriter::row(const std::vector<double> &values, double logpost, int stage)
{
    FILE *f = (FILE *)fp;
    for (double v : values) std::fprintf(f, "%12.6f ", v);
    std::fprintf(f, "%14.6f %5d\n", logpost, stage);
}

const char *param_name(int idx)
{
    static const char *n[B9_NPARAM] = {"logAge", "Y", "FeH", "modulus", "absorption", "carbonicity",
                                       "IFMRconst", "IFMRlin", "IFMRquad", "YB", "lambda", "reserved"};
    return (idx >= 0 && idx < B9_NPARAM) ? n[idx] : "?";
}

}  // namespace b9h
