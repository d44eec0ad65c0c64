#pragma once
#include "hitable.cuh"

// HitableList: the world.  Append order is kept all the way into the kernel because the
// reference resolves equal-distance hits in favour of the earlier entry (hitable_list.cu:18).
// Capacity 1024 as in hitable_list.cuh:10; the flatten kernel reads entries through at().
class HitableList : public Hitable {
 public:
  constexpr static int kMaxHitables = 1024;
  RT_API HitableList() : Hitable(rtapi::H_LIST), count_(0) {}
  RT_API void Append(Hitable *obj) { entries_[count_++] = obj; }
  RT_API int list_len() const { return count_; }
  RT_API Hitable *at(int i) const { return entries_[i]; }

 private:
  int count_;
  Hitable *entries_[kMaxHitables];
};
