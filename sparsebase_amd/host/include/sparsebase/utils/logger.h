// Minimal levelled logger with the reference's interface (utils/logger.h:32-40).
#ifndef SPARSEBASE_UTILS_LOGGER_H_
#define SPARSEBASE_UTILS_LOGGER_H_
#include <iostream>
#include <string>
#include <typeindex>
#include <typeinfo>

#include "sparsebase/utils/utils.h"

namespace sparsebase::utils {
enum LogLevel { LOG_LVL_INFO, LOG_LVL_WARNING, LOG_LVL_NONE };

class Logger {
 public:
  Logger() = default;
  explicit Logger(std::type_index owner) : root_(demangle(owner)) {}
  static void set_level(LogLevel lvl) { level() = lvl; }
  static LogLevel get_level() { return level(); }
  void Log(const std::string &message, LogLevel msg_level = LOG_LVL_INFO) const {
    if (msg_level < level() || msg_level == LOG_LVL_NONE) return;
    std::cerr << "[" << (msg_level == LOG_LVL_WARNING ? "WARNING" : "INFO") << "]"
              << (root_.empty() ? "" : "[" + root_ + "]") << " " << message << std::endl;
  }

 private:
  static LogLevel &level() {
    static LogLevel l = LOG_LVL_WARNING;  // default WARNING, logger.cc:61
    return l;
  }
  std::string root_;
};
}  // namespace sparsebase::utils
#endif
