! sph_hip_textio.f90 -- the token stream behind the hosts' read_data_from_file.
!
! The reference reads its particle files with list-directed reads (SUMMER_SPH.f90:647, "SUMMER_SPH - Variable.f90":782):
! values are separated by blanks, commas or line ends, so a record may continue over line breaks -- and does, in saves
! written with list-directed output under flang, which wraps records at 80 columns.  A plain read(unit,*) of 8 values
! cannot take such a save back (when the 8th value happens to end a line, the value left over on the next line is taken
! for the start of the next record); the hosts therefore walk the token stream themselves and know how long a record is.
module sph_hip_textio
  implicit none
  private
  public :: next_values, tokens_left, load_line
  integer, parameter :: dp = kind(1.0d0)

contains

  logical function is_sep(ch)
    character, intent(in) :: ch
    is_sep = ch == ' ' .or. ch == ',' .or. ch == achar(9) .or. ch == achar(13)
  end function is_sep

  ! the next size(v) values of the stream; nread = how many were found, nlines = lines fetched from the unit on the way
  subroutine next_values(unit_no, line, pos, v, nread, nlines, ios)
    integer, intent(in) :: unit_no
    character(len=*), intent(inout) :: line
    integer, intent(inout) :: pos
    real(dp), intent(out) :: v(:)
    integer, intent(out) :: nread, nlines, ios
    integer :: e, rios
    nread = 0
    nlines = 0
    ios = 0
    do while (nread < size(v))
      do while (pos <= len(line))
        if (.not. is_sep(line(pos:pos))) exit
        pos = pos + 1
      end do
      if (pos > len(line)) then
        read(unit_no, '(A)', iostat=rios) line
        if (rios /= 0) then
          if (nread > 0) ios = rios                ! the file ends inside a record
          return
        end if
        nlines = nlines + 1
        pos = 1
        cycle
      end if
      e = pos
      do while (e <= len(line))
        if (is_sep(line(e:e))) exit
        e = e + 1
      end do
      read(line(pos:e - 1), *, iostat=ios) v(nread + 1)
      if (ios /= 0) return
      nread = nread + 1
      pos = e
    end do
  end subroutine next_values

  ! values between pos and the end of the buffered line
  integer function tokens_left(line, pos)
    character(len=*), intent(in) :: line
    integer, intent(in) :: pos
    integer :: p
    logical :: in_tok
    tokens_left = 0
    in_tok = .false.
    do p = pos, len(line)
      if (is_sep(line(p:p))) then
        in_tok = .false.
      else
        if (.not. in_tok) tokens_left = tokens_left + 1
        in_tok = .true.
      end if
    end do
  end function tokens_left

  ! fetches the next non-blank line into the buffer (pos = 1); ok = .false. at the end of the file
  subroutine load_line(unit_no, line, pos, ok)
    integer, intent(in) :: unit_no
    character(len=*), intent(inout) :: line
    integer, intent(out) :: pos
    logical, intent(out) :: ok
    integer :: rios
    ok = .false.
    do
      read(unit_no, '(A)', iostat=rios) line
      if (rios /= 0) then
        pos = len(line) + 1
        return
      end if
      if (len_trim(line) > 0) exit
    end do
    pos = 1
    ok = .true.
  end subroutine load_line

end module sph_hip_textio
