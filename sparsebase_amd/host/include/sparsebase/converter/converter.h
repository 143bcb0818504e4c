// sparsebase/converter/converter.h — the conversion graph (plug-in point for format
// conversions).  API of the reference's converter/converter.h:72-367; semantics of
// converter.cc:20-288 re-implemented:
//   * two edge maps (copy / move): from-type -> to-type -> [(condition, function)];
//   * GetConversionChain: breadth-first search over format types, unit edge cost; an
//     edge is usable if its condition accepts (source context, one of the target
//     contexts) — evaluated with the ORIGINAL source context on every hop, as the
//     reference does (converter.cc:163);
//   * Convert deletes intermediates, ConvertCached returns them.
#ifndef SPARSEBASE_CONVERTER_CONVERTER_H_
#define SPARSEBASE_CONVERTER_CONVERTER_H_
#include <algorithm>
#include <deque>
#include <functional>
#include <memory>
#include <optional>
#include <tuple>
#include <typeindex>
#include <unordered_map>
#include <vector>

#include "sparsebase/config.h"
#include "sparsebase/utils/utils.h"

namespace sparsebase {
namespace format {
class Format;
}
namespace context {
struct Context;
}
namespace converter {

typedef std::function<format::Format *(format::Format *, context::Context *)> ConversionFunction;
typedef std::function<bool(context::Context *, context::Context *)> ConversionCondition;
typedef std::tuple<ConversionFunction, context::Context *, utils::CostType> ConversionStep;
typedef std::optional<std::tuple<std::vector<ConversionStep>, utils::CostType>> ConversionChain;
typedef std::vector<ConversionChain> ConversionSchema;
typedef std::unordered_map<
    std::type_index,
    std::unordered_map<std::type_index, std::vector<std::tuple<ConversionCondition, ConversionFunction>>>>
    ConversionMap;

class Converter {
 public:
  void RegisterConversionFunction(std::type_index from_type, std::type_index to_type, ConversionFunction conv_func,
                                  ConversionCondition edge_condition, bool is_move_conversion = false) {
    (*map_for(is_move_conversion))[from_type][to_type].emplace_back(std::move(edge_condition), std::move(conv_func));
  }

  format::Format *Convert(format::Format *source, std::type_index to_type, context::Context *to_context,
                          bool is_move_conversion = false) const {
    return Convert(source, to_type, std::vector<context::Context *>{to_context}, is_move_conversion);
  }
  format::Format *Convert(format::Format *source, std::type_index to_type,
                          std::vector<context::Context *> to_contexts, bool is_move_conversion = false) const;
  std::vector<format::Format *> ConvertCached(format::Format *source, std::type_index to_type,
                                              context::Context *to_context, bool is_move_conversion = false) const {
    return ConvertCached(source, to_type, std::vector<context::Context *>{to_context}, is_move_conversion);
  }
  std::vector<format::Format *> ConvertCached(format::Format *source, std::type_index to_type,
                                              std::vector<context::Context *> to_contexts,
                                              bool is_move_conversion = false) const;

  template <typename FormatType>
  FormatType *Convert(format::Format *source, context::Context *to_context, bool is_move_conversion = false) const;
  template <typename FormatType>
  FormatType *Convert(format::Format *source, std::vector<context::Context *> to_contexts,
                      bool is_move_conversion = false) const;

  ConversionChain GetConversionChain(std::type_index from_type, context::Context *from_context,
                                     std::type_index to_type, const std::vector<context::Context *> &to_contexts,
                                     bool is_move_conversion = false) const {
    if (from_type == to_type &&
        std::find(to_contexts.begin(), to_contexts.end(), from_context) != to_contexts.end())
      return ConversionChain(std::in_place);  // nothing to do, but possible
    auto steps = Search(from_type, from_context, to_type, to_contexts, map_for(is_move_conversion));
    if (steps.empty()) return {};
    const utils::CostType cost = static_cast<utils::CostType>(steps.size());
    return std::make_tuple(std::move(steps), cost);
  }

  bool CanConvert(std::type_index from_type, context::Context *from_context, std::type_index to_type,
                  context::Context *to_context, bool is_move_conversion = false) const {
    return GetConversionChain(from_type, from_context, to_type, {to_context}, is_move_conversion).has_value();
  }
  bool CanConvert(std::type_index from_type, context::Context *from_context, std::type_index to_type,
                  const std::vector<context::Context *> &to_contexts, bool is_move_conversion = false) const {
    return GetConversionChain(from_type, from_context, to_type, to_contexts, is_move_conversion).has_value();
  }

  void ClearConversionFunctions(std::type_index from_type, std::type_index to_type, bool move_conversion = false) {
    auto *map = map_for(move_conversion);
    auto it = map->find(from_type);
    if (it == map->end()) return;
    it->second.erase(to_type);
    if (it->second.empty()) map->erase(it);
  }
  void ClearConversionFunctions(bool move_conversion = false) { map_for(move_conversion)->clear(); }

  // chain = [input, intermediates..., result]; with clear_intermediate the
  // intermediates are deleted on the way and only [input, result] is returned
  static std::vector<format::Format *> ApplyConversionChain(const ConversionChain &chain, format::Format *input,
                                                            bool clear_intermediate);
  static std::vector<std::vector<format::Format *>> ApplyConversionSchema(
      const ConversionSchema &cs, const std::vector<format::Format *> &packed_sfs, bool clear_intermediate) {
    std::vector<std::vector<format::Format *>> out;
    for (size_t i = 0; i < cs.size(); i++) out.push_back(ApplyConversionChain(cs[i], packed_sfs[i], clear_intermediate));
    return out;
  }

  virtual std::type_index get_converter_type() const = 0;
  virtual Converter *Clone() const = 0;
  virtual void Reset() = 0;
  virtual ~Converter() = default;

 private:
  ConversionMap copy_map_, move_map_;
  ConversionMap *map_for(bool move) { return move ? &move_map_ : &copy_map_; }
  const ConversionMap *map_for(bool move) const { return move ? &move_map_ : &copy_map_; }

  static std::vector<ConversionStep> Search(std::type_index from_type, context::Context *from_context,
                                            std::type_index to_type,
                                            const std::vector<context::Context *> &to_contexts,
                                            const ConversionMap *map) {
    struct Visit {
      std::type_index prev;
      ConversionStep step;
    };
    std::unordered_map<std::type_index, Visit> seen;
    seen.emplace(from_type, Visit{from_type, ConversionStep(nullptr, nullptr, 0)});
    std::deque<std::type_index> frontier{from_type};
    while (!frontier.empty()) {
      const std::type_index cur = frontier.front();
      frontier.pop_front();
      auto edges = map->find(cur);
      if (edges == map->end()) continue;
      for (const auto &target : edges->second) {
        if (seen.count(target.first)) continue;
        bool taken = false;
        for (const auto &cond_fn : target.second) {
          for (auto *ctx : to_contexts) {
            if (std::get<0>(cond_fn)(from_context, ctx)) {
              seen.emplace(target.first, Visit{cur, ConversionStep(std::get<1>(cond_fn), ctx, 1)});
              frontier.push_back(target.first);
              taken = true;
              break;
            }
          }
          if (taken) break;
        }
        if (taken && target.first == to_type) {
          std::vector<ConversionStep> path;
          for (std::type_index t = to_type; t != from_type; t = seen.at(t).prev) path.push_back(seen.at(t).step);
          std::reverse(path.begin(), path.end());
          return path;
        }
      }
    }
    return {};
  }
};

template <class ConverterType>
class ConverterImpl : public Converter {
 public:
  std::type_index get_converter_type() const override { return typeid(ConverterType); }
};

}  // namespace converter
}  // namespace sparsebase
#endif
