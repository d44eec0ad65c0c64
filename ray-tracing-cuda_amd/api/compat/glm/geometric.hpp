#pragma once
#include "glm.hpp"
