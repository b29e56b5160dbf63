#pragma once
#include <boost/geometry.hpp>
