#pragma once
#include <exception>
#include <string>
namespace OpenMM {
class OpenMMException : public std::exception {
 public:
  explicit OpenMMException(const std::string& message) : message(message) {}
  ~OpenMMException() noexcept override {}
  const char* what() const noexcept override { return message.c_str(); }

 private:
  std::string message;
};
}  // namespace OpenMM
