// enum_names.hpp -- helper of the header mirror: builds the `<enum>_name` look-up tables that the reference declares next to
// each option enum (reference: lgrngn/kernel.hpp:11-24, terminal_velocity.hpp:10-17, advection_scheme.hpp:10-15, RH_formula.hpp:14-19,
// ccn_source.hpp:14-18, backend.hpp:10-16) from ONE list of enumerator names in declaration order, so that a name cannot drift
// from its enumerator.  Same type as the reference's tables (std::unordered_map<E, std::string>): `kernel_name.at(k)`,
// range-for over the table etc. of a driver that prints its options compile unchanged.
#pragma once
#include <initializer_list>
#include <string>
#include <unordered_map>
namespace libcloudphxx { namespace lgrngn { namespace detail {
  template <class E>
  inline std::unordered_map<E, std::string> enum_names(std::initializer_list<const char *> names_in_declaration_order)
  {
    std::unordered_map<E, std::string> m;
    int v = 0;
    for (const char *nm : names_in_declaration_order) m.emplace(static_cast<E>(v++), nm);
    return m;
  }
} } }
