// sparsebase/utils/function_matcher_mixin.h — the operator plug-in point: a map
// from input-format type keys to implementation functions, with automatic input
// conversion.  API and dispatch rules of the reference's
// utils/function_matcher_mixin.h:35-418, re-implemented:
//   * direct hit  : key registered AND every input lives in one of `contexts`;
//   * otherwise   : every registered key of the same arity is costed — 0 for an
//                   input whose type already equals the key's (no context check, as in
//                   the reference :366-368), else the length of the conversion chain —
//                   and the cheapest usable key wins; none usable -> FunctionNotFoundException;
//   * convert_input == false with a non-empty chain -> DirectExecutionNotAvailableException;
//   * Execute deletes the formats created by conversion, CachedExecute returns them.
#ifndef SPARSEBASE_UTILS_FUNCTION_MATCHER_MIXIN_H_
#define SPARSEBASE_UTILS_FUNCTION_MATCHER_MIXIN_H_
#include <limits>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "sparsebase/converter/converter.h"
#include "sparsebase/format/format.h"
#include "sparsebase/utils/parameterizable.h"
#include "sparsebase/utils/utils.h"

namespace sparsebase::utils {

template <typename ReturnType>
using PreprocessFunction = ReturnType (*)(std::vector<format::Format *> formats, utils::Parameters *params);

template <typename ReturnType, class PreprocessingImpl = Parameterizable,
          typename Function = PreprocessFunction<ReturnType>, typename Key = std::vector<std::type_index>,
          typename KeyHash = TypeIndexVectorHash, typename KeyEqualTo = std::equal_to<std::vector<std::type_index>>>
class FunctionMatcherMixin : public PreprocessingImpl {
  typedef std::unordered_map<Key, Function, KeyHash, KeyEqualTo> ConversionMap;

 public:
  std::vector<Key> GetAvailableFormats() {
    std::vector<Key> keys;
    for (const auto &kv : map_to_function_) keys.push_back(kv.first);
    return keys;
  }
  bool RegisterFunctionNoOverride(const Key &key, const Function &fn) {
    return map_to_function_.emplace(key, fn).second;
  }
  void RegisterFunction(const Key &key, const Function &fn) { map_to_function_[key] = fn; }
  bool UnregisterFunction(const Key &key) { return map_to_function_.erase(key) > 0; }

 protected:
  using PreprocessingImpl::PreprocessingImpl;
  ConversionMap map_to_function_;

  bool CheckIfKeyMatches(const ConversionMap &map, const Key &key, const std::vector<format::Format *> &formats,
                         const std::vector<context::Context *> &contexts) {
    if (map.find(key) == map.end()) return false;
    for (auto *f : formats) {
      bool placed = false;
      for (auto *c : contexts) placed = placed || f->get_context()->IsEquivalent(c);
      if (!placed) return false;
    }
    return true;
  }

  std::tuple<Function, converter::ConversionSchema> GetFunction(const std::vector<format::Format *> &formats,
                                                                const Key &key, const ConversionMap &map,
                                                                const std::vector<context::Context *> &contexts) {
    if (CheckIfKeyMatches(map, key, formats, contexts))
      return std::make_tuple(map.at(key), converter::ConversionSchema(key.size()));
    Function best_fn = nullptr;
    converter::ConversionSchema best_schema;
    unsigned best_cost = std::numeric_limits<unsigned>::max();
    for (const auto &candidate : map) {
      const Key &ck = candidate.first;
      if (ck.size() != key.size()) continue;
      converter::ConversionSchema schema;
      unsigned cost = 0;
      bool usable = true;
      for (size_t i = 0; i < ck.size() && usable; i++) {
        if (key[i] == ck[i]) {
          schema.push_back({});
          continue;
        }
        auto chain = formats[i]->get_converter()->GetConversionChain(key[i], formats[i]->get_context(), ck[i],
                                                                      contexts);
        if (!chain) {
          usable = false;
        } else {
          cost += std::get<1>(*chain);
          schema.push_back(*chain);
        }
      }
      if (usable && cost < best_cost) {
        best_cost = cost;
        best_fn = candidate.second;
        best_schema = schema;
      }
    }
    if (best_fn == nullptr) {
      std::string msg = "Could not find a function that matches the formats: {";
      for (auto *f : formats) msg += f->get_name() + " ";
      msg += "} using the contexts {";
      for (auto *c : contexts) msg += c->get_name() + " ";
      msg += "}";
      throw FunctionNotFoundException(msg);
    }
    return std::make_tuple(best_fn, best_schema);
  }

  template <typename F, typename... SF>
  std::tuple<std::vector<std::vector<format::Format *>>, ReturnType> CachedExecute(
      utils::Parameters *params, std::vector<context::Context *> contexts, bool convert_input,
      bool clear_intermediate, F format, SF... formats) {
    std::vector<format::Format *> packed{format, formats...};
    Key key;
    for (auto *f : packed) key.push_back(f->get_id());
    auto fn_schema = GetFunction(packed, key, map_to_function_, contexts);
    Function fn = std::get<0>(fn_schema);
    const converter::ConversionSchema &schema = std::get<1>(fn_schema);
    if (!convert_input)
      for (const auto &chain : schema)
        if (chain) throw DirectExecutionNotAvailableException<Key>(key, this->GetAvailableFormats());
    auto chains = converter::Converter::ApplyConversionSchema(schema, packed, clear_intermediate);
    std::vector<format::Format *> inputs;
    std::vector<std::vector<format::Format *>> created;
    for (auto &c : chains) {
      inputs.push_back(c.back());
      created.emplace_back(c.begin() + 1, c.end());
    }
    return std::make_tuple(created, fn(inputs, params));
  }

  template <typename F, typename... SF>
  ReturnType Execute(utils::Parameters *params, std::vector<context::Context *> contexts, bool convert_input, F sf,
                     SF... sfs) {
    auto out = CachedExecute(params, contexts, convert_input, true, sf, sfs...);
    for (auto &chain : std::get<0>(out))
      for (auto *f : chain) delete f;
    return std::get<1>(out);
  }
};

}  // namespace sparsebase::utils
#endif
