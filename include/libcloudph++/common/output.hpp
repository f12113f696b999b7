// Keys of diag_puddle(): same enumerators and order as the reference's common::output_t
// (reference: include/libcloudph++/common/output.hpp:8-42) == enum lcx_puddle in include/lcx.h.
#pragma once
#include <map>
#include <stdexcept>
#include <string>
namespace libcloudphxx { namespace common {
  namespace chem { enum chem_species_t { HNO3, NH3, CO2, SO2, H2O2, O3, S_VI, H, chem_all = H + 1, chem_gas_n = O3 + 1 }; }
  enum output_t { outHNO3, outNH3, outCO2, outSO2, outH2O2, outO3, outS_VI, outH,
                  outliq_vol, outdry_vol, outprtcl_num, outice_mass, outliq_num, outice_num };
  inline const std::map<output_t, std::string> &output_name_map()
  {
    static const std::map<output_t, std::string> m = {
      {outHNO3, "HNO3"}, {outNH3, "NH3"}, {outCO2, "CO2"}, {outSO2, "SO2"}, {outH2O2, "H2O2"}, {outO3, "O3"}, {outS_VI, "S_VI"}, {outH, "H"},
      {outliq_vol, "liquid_volume"}, {outdry_vol, "dry_volume"}, {outprtcl_num, "particle_number"}, {outice_mass, "ice_mass"},
      {outliq_num, "liquid_number"}, {outice_num, "ice_number"}};
    return m;
  }
  static const std::map<output_t, std::string> &output_names = output_name_map();
  inline output_t get_output_enum(const std::string &name)
  {
    for (const auto &kv : output_name_map()) if (kv.second == name) return kv.first;
    throw std::runtime_error("Incorrect name for puddle: " + name);
  }
} }
